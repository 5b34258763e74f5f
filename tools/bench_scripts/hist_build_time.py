"""The colour-histogram build of an 8K image (dp_kmeans_hist_build_u8: count / plan / zero / scatter / parts kernels): events around the
call, noise / image-like / flat content.  Under rocprofv3 --kernel-trace --stats the per-kernel split.  usage: hist_build_time.py [reps]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dither_pie_amd import backend
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
h, w = 4320, 7680
imgs = {"noise": torch.from_numpy(np.random.RandomState(99).randint(0, 256, (h, w, 3), dtype=np.uint8)).cuda()}
y, x = np.mgrid[0:h, 0:w]
a = np.stack([80 + 60 * np.sin(x / 300.0) + 40 * (y / h), 110 + 50 * np.cos(y / 200.0) + 20 * np.sin(x / 97.0), 160 + 70 * (y / h) + 10 * np.sin((x + y) / 50.0)], -1)
imgs["image-like"] = torch.from_numpy(np.clip(a + np.random.RandomState(3).normal(0, 3, a.shape), 0, 255).astype(np.uint8)).cuda()
imgs["flat"] = torch.full((h, w, 3), 77, dtype=torch.uint8, device="cuda")
for name, img in imgs.items():
    flat = img.reshape(-1, 3)
    hist = backend.ColourHistogram(flat); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); hist.add(flat, False); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    # the table against torch's own count of the packed colours (exact)
    packed = (flat[:, 0].to(torch.int64) << 16) | (flat[:, 1].to(torch.int64) << 8) | flat[:, 2].to(torch.int64)
    cm = ((packed >> 20) & 15) << 20 | ((packed >> 12) & 15) << 16 | ((packed >> 4) & 15) << 12 | ((packed >> 16) & 15) << 8 | ((packed >> 8) & 15) << 4 | (packed & 15)
    ref = torch.bincount(cm, minlength=1 << 24).to(torch.int32)
    tab = hist.buf[: 1 << 26].view(torch.int32)
    print(f"{name:11s} min {ts[0]:.4f} median {ts[len(ts) // 2]:.4f} ms   table exact: {bool(torch.equal(tab, ref))}", flush=True)
    del hist
