"""Soak of dp_distinct_first_u8 (the distinct colours of an image in order of first occurrence) and of the colour histogram
(dp_kmeans_hist_build_u8, incl. accumulation) against numpy: random pixel counts around the kernels' block / tile / part sizes,
contents from noise to runs to a handful of colours, aligned and unaligned buffers.  usage: fuzz_distinct.py [seed] [cases]"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import backend as be


def first_occurrences(arr):
    packed = (arr[:, 0].astype(np.uint32) << 16) | (arr[:, 1].astype(np.uint32) << 8) | arr[:, 2]
    _, first = np.unique(packed, return_index=True)
    return arr[np.sort(first)]


def hist_table(px):
    r, g, b = (px[:, i].astype(np.int64) for i in range(3))
    idx = ((r >> 4) << 20) | ((g >> 4) << 16) | ((b >> 4) << 12) | ((r & 15) << 8) | ((g & 15) << 4) | (b & 15)
    return np.bincount(idx, minlength=1 << 24).astype(np.uint32)


def run(seed, cases):
    rs = np.random.RandomState(seed)
    bad = 0
    t0 = time.time()
    for case in range(cases):
        n = int(rs.choice([1, 2, 5, 63, 64, 255, 256, 2047, 2048, 2049, 4095, 8191, 8192, 8193, 16383, 16384, 16385, 65536, 70001, 300007, 1_000_003]))
        kind = int(rs.randint(0, 6))
        if kind == 0:
            px = rs.randint(0, 256, (n, 3)).astype(np.uint8)
        elif kind == 1:
            px = rs.randint(0, 256, (int(rs.randint(1, 9)), 3)).astype(np.uint8)[rs.randint(0, 8, n) % 1 + rs.randint(0, 1, n)]
        elif kind == 2:
            run_len = int(rs.choice([2, 3, 4, 5, 37, 1000]))
            px = np.repeat(rs.randint(0, 256, (n // run_len + 1, 3)).astype(np.uint8), run_len, axis=0)[:n]
        elif kind == 3:
            px = np.tile(rs.randint(0, 256, (1, 3)).astype(np.uint8), (n, 1))
        elif kind == 4:   # everything inside one or two cells of the colour cube (crowded buckets, many parts)
            base = rs.randint(0, 240, 3)
            px = (base + rs.randint(0, 18, (n, 3))).astype(np.uint8)
        else:             # a ramp with grain
            t = np.arange(n)
            px = np.clip(np.stack([t * 255 // max(n - 1, 1), 255 - t * 255 // max(n - 1, 1), (t // 7) % 256], -1) + rs.randint(-2, 3, (n, 3)), 0, 255).astype(np.uint8)
        px = np.ascontiguousarray(px)
        off = int(rs.choice([0, 0, 1, 2, 3]))
        raw = torch.empty(3 * n + 3, dtype=torch.uint8, device="cuda")
        raw[off:off + 3 * n] = torch.from_numpy(px).cuda().reshape(-1)
        t = raw[off:off + 3 * n].view(n, 3)
        ok = np.array_equal(be.distinct_first(t).cpu().numpy(), first_occurrences(px))
        hist = be.ColourHistogram(t)
        ref = hist_table(px)
        ok = ok and np.array_equal(hist.buf[:1 << 26].view(torch.int32).cpu().numpy().view(np.uint32), ref)
        if n > 10 and rs.rand() < 0.5:   # accumulate a second buffer on top
            m = int(rs.randint(1, n))
            hist.add(t[:m], accumulate=True)
            ok = ok and np.array_equal(hist.buf[:1 << 26].view(torch.int32).cpu().numpy().view(np.uint32), ref + hist_table(px[:m]))
            info = hist.buf[1 << 26:].view(torch.int32).cpu().numpy().view(np.uint32)
            per_cell = (ref + hist_table(px[:m])).reshape(4096, 4096).sum(1).astype(np.uint32)
            occ = np.nonzero(per_cell)[0]
            ok = ok and np.array_equal(info[:4096], per_cell) and info[4096] == len(occ) and np.array_equal(info[4097:4097 + len(occ)] & 0xfff, occ)
        if not ok:
            bad += 1
            print("MISMATCH", seed, case, n, kind, off, flush=True)
    print(f"fuzz_distinct: {cases} cases, {bad} mismatching, {time.time() - t0:.1f} s")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 100) else 0)
