"""Soak of the Lloyd passes against the oracle's float64 labelling: the case generator of
tests/test_gpu_fullsize.py::test_kmeans_cell_list_fuzz over many seeds (run on the GPU box).
usage: fuzz_kmeans.py <first seed> <seeds> [cells|hist]
  cells (default): the candidate-list pass over the pixels (dp_kmeans_step_u8 with DP_KMEANS_CELLS=1; the experiment library)
  hist:            the pass over the colour histogram (dp_kmeans_hist_build_u8 + dp_kmeans_hist_step; the product library) --
                   the histogram is built by partition, every pass builds its candidate lists in-kernel"""
import os, sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from oracle import oracle as orc
from dither_pie_amd import backend as be


def run(seed0, seeds, which="cells"):
    orc.build()
    if which == "cells":
        os.environ["DP_KMEANS_CELLS"] = "1"
    bad = cases = 0
    t0 = time.time()
    for seed in range(seed0, seed0 + seeds):
        rs = np.random.RandomState(1000 + seed)
        for case in range(12):
            K = int(rs.choice([1, 2, 3, 7, 16, 31, 32, 64, 65, 129, 255, 256]))
            n = int(rs.choice([1, 3, 255, 257, 4096, 50001, 200003]))
            kind = rs.randint(0, 4)
            if kind == 0:
                px = rs.randint(0, 256, (n, 3)).astype(np.uint8)
            elif kind == 1:
                t = np.arange(n)
                px = np.clip(np.stack([t * 255 // max(n - 1, 1), 255 - t * 255 // max(n - 1, 1), (t // 7) % 256], -1) + rs.randint(-2, 3, (n, 3)), 0, 255).astype(np.uint8)
            elif kind == 2:
                px = rs.randint(0, 256, (5, 3)).astype(np.uint8)[rs.randint(0, 5, n)]
            else:
                px = rs.randint(0, 40, (n, 3)).astype(np.uint8)
            ckind = rs.randint(0, 4)
            if ckind == 0:
                centers = rs.rand(K, 3) * 255.0
            elif ckind == 1:
                centers = np.round(rs.rand(K, 3) * 255.0)
            elif ckind == 2:
                centers = px[rs.randint(0, n, K)].astype(np.float64) + rs.choice([0.0, 0.5, 0.25])
            else:
                centers = 5.0 + rs.rand(K, 3) * 20.0
            # half of the cases with sklearn's tie rule (mean_dev: labels of equidistant pixels from sklearn's float64 expression)
            mean = orc.data_mean(px) if rs.rand() < 0.5 else None
            s_ref, n_ref, _ = orc.kmeans_step(px, centers, mean)
            if which == "hist":
                s, cnt, _q = be.ColourHistogram(torch.from_numpy(px).cuda()).step(torch.from_numpy(centers), None if mean is None else torch.from_numpy(mean))
            else:
                s, cnt, _q = be.kmeans_step(torch.from_numpy(px).cuda(), torch.from_numpy(centers), None if mean is None else torch.from_numpy(mean))
            cases += 1
            if not (np.array_equal(s.cpu().numpy(), s_ref) and np.array_equal(cnt.cpu().numpy(), n_ref)):
                bad += 1
                print("MISMATCH", seed, case, K, n, kind, ckind, flush=True)
        if (seed - seed0) % 20 == 19:
            print(f"  {seed - seed0 + 1} seeds, {cases} cases, {bad} mismatching, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz_kmeans ({which}): {cases} cases, {bad} mismatching, {time.time() - t0:.1f} s")
    return bad


if __name__ == "__main__":
    from dither_pie_amd import _lib
    which = sys.argv[3] if len(sys.argv) > 3 else "cells"
    if which == "cells" and not _lib.EXPERIMENTS: _lib.select(True)   # DP_KMEANS_CELLS is read by libditherpie_hip_exp.so only
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 50, which) else 0)
