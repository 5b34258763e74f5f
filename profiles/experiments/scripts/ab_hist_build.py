"""A/B of two BUILDS (ab_libs.py) for the colour-histogram build of an 8K image: noise and image-like content.  usage: ab_hist_build.py"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import _lib, backend, dithering_lib
h, w = 4320, 7680
imgs = {"noise": torch.from_numpy(np.random.RandomState(99).randint(0, 256, (h, w, 3), dtype=np.uint8)).cuda()}
y, x = np.mgrid[0:h, 0:w]
a = np.stack([80 + 60 * np.sin(x / 300.0) + 40 * (y / h), 110 + 50 * np.cos(y / 200.0) + 20 * np.sin(x / 97.0), 160 + 70 * (y / h) + 10 * np.sin((x + y) / 50.0)], -1)
imgs["image-like"] = torch.from_numpy(np.clip(a + np.random.RandomState(3).normal(0, 3, a.shape), 0, 255).astype(np.uint8)).cuda()
for name, img in imgs.items():
    flat = img.reshape(-1, 3)
    best = {False: [], True: []}
    for rep in range(3):
        for exp in (False, True):
            _lib.select(exp); dithering_lib.drop_device_caches()
            hist = backend.ColourHistogram(flat); torch.cuda.synchronize()
            ts = []
            for _ in range(8):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); hist.add(flat, False); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            best[exp].append(min(ts)); del hist
    a_, b_ = min(best[False]), min(best[True])
    print(f"{name:11s} A (product) {a_:.4f} ms   B (experiments twin) {b_:.4f} ms   B/A {b_ / a_:.3f}   [A {' '.join('%.4f' % v for v in best[False])} | B {' '.join('%.4f' % v for v in best[True])}]", flush=True)
