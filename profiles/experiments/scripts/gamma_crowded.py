"""use_gamma with a palette extracted from the content (median cut of the linearised image, as apply_dithering does with
palette=None): ordered_compact_float_kernel against ordered_lean_float_kernel (DP_NO_COMPACT_KERNEL=1), 24 4K frames.
usage: gamma_crowded.py [SWITCH]   (another DP_* switch of the experiments library for the second leg)"""
import os, sys; sys.path.insert(0, '.')
os.environ["DITHER_PIE_EXPERIMENTS"] = "1"
import numpy as np, torch
from PIL import Image
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
from dither_pie_amd import _tables
rs = np.random.RandomState(3)
h, w = 540, 960
y, x = np.mgrid[0:h, 0:w]
def img(kind):
    if kind == "smooth":
        r = 80 + 60 * np.sin(x / 300.0) + 40 * (y / h); g = 110 + 50 * np.cos(y / 200.0) + 20 * np.sin(x / 97.0); b = 160 + 70 * (y / h) + 10 * np.sin((x + y) / 50.0)
    else:
        r = 20 + 25 * np.sin(x / 120.0) ** 2 + 15 * (y / h); g = 18 + 22 * np.cos(y / 90.0) ** 2; b = 25 + 30 * np.sin((x + y) / 150.0) ** 2
    return np.clip(np.stack([r, g, b], -1) + rs.normal(0, 3, (h, w, 3)), 0, 255).astype(np.uint8)
def timeit(fn, n=5):
    for _ in range(3): fn()
    ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
for kind in ("smooth", "dark"):
    a = img(kind)
    f = torch.from_numpy(a).cuda().repeat(4, 4, 1).unsqueeze(0).repeat(24, 1, 1, 1).contiguous()
    outs = [torch.empty_like(f) for _ in range(2)]
    for K in (256, 64):
        pal = ColorReducer.reduce_colors(Image.fromarray(_tables.LUT_IN[a], "RGB"), K)   # dithering_lib.py:1960-1966
        d = ImageDitherer(K, DitherMode.BAYER, pal, True, {"size": "8x8"}).prepare()
        res = []
        for i, env in enumerate(({}, {(sys.argv[1] if len(sys.argv) > 1 else "DP_NO_COMPACT_KERNEL"): "1"})):
            for k, v in env.items(): os.environ[k] = v
            res.append(timeit(lambda: d.apply_dithering_frames(f, out=outs[i])))
            for k in env: del os.environ[k]
        print(f"gamma + median cut {K:3d} of the {kind} content, bayer8: compact float {res[0]:.3f} ms ({24*2160*3840/res[0]/1e6:.1f} Gpx/s) | "
              f"lean float {res[1]:.3f} ms | identical: {torch.equal(outs[0], outs[1])}", flush=True)
