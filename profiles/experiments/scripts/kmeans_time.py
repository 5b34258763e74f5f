"""Lloyd pass over the C4 image (33 M pixels), event-timed, VALU kernel vs matrix-core kernel; totals compared.
usage: kmeans_time.py [K ...]"""
import os; os.environ.setdefault("DITHER_PIE_EXPERIMENTS", "1")  # the DP_* switches live in libditherpie_hip_exp.so
import sys, os; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import backend as be
g = torch.Generator(device='cuda'); g.manual_seed(99)
px = torch.randint(0, 256, (4320 * 7680, 3), dtype=torch.uint8, device='cuda', generator=g)
for K in ([int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128, 256]):
    c = torch.from_numpy(np.random.RandomState(1).rand(K, 3) * 255.0).cuda()
    res = {}
    for which in ("VALU", "MFMA"):
        os.environ.pop("DP_KMEANS_MFMA", None)
        if which == "MFMA": os.environ["DP_KMEANS_MFMA"] = "1"
        tot = torch.zeros(5 * K, dtype=torch.int64, device='cuda')
        for _ in range(2): be.kmeans_step_into(px, c, tot, want_sq=True)
        ts = []
        for _ in range(6):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); be.kmeans_step_into(px, c, tot, want_sq=False); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[which] = (min(ts), tot.cpu().numpy().copy())
    same = bool((res["VALU"][1][:4 * K] == res["MFMA"][1][:4 * K]).all())
    print(f"K={K:3d}: VALU {res['VALU'][0]:.4f} ms   MFMA {res['MFMA'][0]:.4f} ms   (incl. the memsets)  totals equal: {same}", flush=True)
