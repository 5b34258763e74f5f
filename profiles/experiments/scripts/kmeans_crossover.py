"""Where the candidate-list Lloyd pass overtakes the full scan: event-timed passes (incl. memset and list build) over
n pixels of uniform noise, both paths forced.  usage: kmeans_crossover.py"""
import os; os.environ.setdefault("DITHER_PIE_EXPERIMENTS", "1")  # the DP_* switches live in libditherpie_hip_exp.so
import sys, os; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import backend as be
g = torch.Generator(device='cuda'); g.manual_seed(5)
big = torch.randint(0, 256, (1 << 24, 3), dtype=torch.uint8, device='cuda', generator=g)
for K in (4, 8, 16, 32, 64, 128, 256):
    c = torch.from_numpy(np.random.RandomState(1).rand(K, 3) * 255.0).cuda()
    os.environ["DP_KMEANS_CELLS"] = "0"
    for _ in range(3):
        s_, n_, _q = be.kmeans_step(big[::64].contiguous(), c)
        c = torch.where(n_[:, None] > 0, s_.double() / n_.clamp(min=1)[:, None].double(), c)
    row = []
    for lg in (18, 19, 20, 21, 22, 23, 24):
        px = big[: 1 << lg]
        t = {}
        for which in ("0", "1"):
            os.environ["DP_KMEANS_CELLS"] = which
            tot = torch.zeros(5 * K, dtype=torch.int64, device='cuda')
            for _ in range(3): be.kmeans_step_into(px, c, tot, want_sq=False)
            ts = []
            for _ in range(10):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); be.kmeans_step_into(px, c, tot, want_sq=False); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            t[which] = sorted(ts)[2]
        row.append(f"2^{lg}: {t['0']:6.1f}/{t['1']:6.1f}")
    print(f"K={K:3d} scan/cells us  " + "  ".join(row), flush=True)
os.environ.pop("DP_KMEANS_CELLS", None)
