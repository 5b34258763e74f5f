"""Lloyd pass over the C4 image (33 M pixels), event-timed: full scan (DP_KMEANS_CELLS=0) vs per-cell candidate lists;
totals compared.  Also a smooth image-like input and centres that are real k-means centres of the data.
usage: kmeans_cells_time.py [K ...]"""
import os; os.environ.setdefault("DITHER_PIE_EXPERIMENTS", "1")  # the DP_* switches live in libditherpie_hip_exp.so
import sys, os; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import backend as be
g = torch.Generator(device='cuda'); g.manual_seed(99)
N = 4320 * 7680
rnd = torch.randint(0, 256, (N, 3), dtype=torch.uint8, device='cuda', generator=g)
yy, xx = torch.meshgrid(torch.arange(4320, device='cuda'), torch.arange(7680, device='cuda'), indexing='ij')
smooth = torch.stack([(xx * 255 // 7679), (yy * 255 // 4319), ((xx + yy) * 255 // (7679 + 4319))], -1).to(torch.uint8).reshape(-1, 3)
smooth = (smooth.to(torch.int16) + torch.randint(-6, 7, smooth.shape, device='cuda', generator=g).to(torch.int16)).clamp(0, 255).to(torch.uint8).contiguous()
for name, px in (("uniform random", rnd), ("smooth + grain", smooth)):
    for K in ([int(a) for a in sys.argv[1:]] or [1, 2, 8, 16, 32, 64, 128, 256]):
        c = torch.from_numpy(np.random.RandomState(1).rand(K, 3) * 255.0).cuda()
        # a few Lloyd steps so that the centres look like k-means centres of this data
        for _ in range(3):
            s, n_, _q = be.kmeans_step(px[::64].contiguous(), c)
            c = torch.where(n_[:, None] > 0, s.double() / n_.clamp(min=1)[:, None].double(), c)
        res = {}
        for which in ("scan", "cells"):
            os.environ["DP_KMEANS_CELLS"] = "0" if which == "scan" else "1"
            tot = torch.zeros(5 * K, dtype=torch.int64, device='cuda')
            for _ in range(2): be.kmeans_step_into(px, c, tot, want_sq=True)
            sq = tot.cpu().numpy().copy()
            ts = []
            for _ in range(6):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); be.kmeans_step_into(px, c, tot, want_sq=False); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            res[which] = (min(ts), tot.cpu().numpy().copy(), sq)
        same = bool((res["scan"][1][:4 * K] == res["cells"][1][:4 * K]).all()) and bool((res["scan"][2] == res["cells"][2]).all())
        print(f"{name:15s} K={K:3d}: scan {res['scan'][0]:.4f} ms   cells {res['cells'][0]:.4f} ms   (incl. memset + table build)  totals equal: {same}", flush=True)
os.environ.pop("DP_KMEANS_CELLS", None)
