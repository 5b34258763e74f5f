"""Crowded palettes (median cut / k-means of the content itself), 24 4K frames: the compact kernel (default) against the
adaptive lean kernel (DP_NO_COMPACT_KERNEL=1) and against one workgroup per CU (DP_COMPACT_NO_HALF=1), same process, same
box; outputs compared byte for byte.  usage: crowded_ab.py [K ...] [DP_SWITCH]   (a DP_* switch of the experiments library replaces the
third leg: how the variants of profiles/experiments/r04_compact_kernel_trials.md were measured)"""
import os, sys; sys.path.insert(0, '.')
os.environ["DITHER_PIE_EXPERIMENTS"] = "1"
import numpy as np, torch
from PIL import Image
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
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
    fn(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
Ks = [int(v) for v in sys.argv[1:] if v.isdigit()] or [256, 64]
THIRD = ([v for v in sys.argv[1:] if v.startswith("DP_")] or ["DP_COMPACT_NO_HALF"])[0]
for kind in ("smooth", "dark"):
    a = img(kind)
    f = torch.from_numpy(a).cuda().repeat(4, 4, 1).unsqueeze(0).repeat(24, 1, 1, 1).contiguous()
    outs = [torch.empty_like(f) for _ in range(3)]
    for K in Ks:
        for src, pal in (("median_cut", ColorReducer.reduce_colors(Image.fromarray(a, "RGB"), K)),
                         ("kmeans", ColorReducer.generate_kmeans_palette(Image.fromarray(a, "RGB"), K, random_state=42))):
            for mode, params in ((DitherMode.BAYER, {"size": "8x8"}), (DitherMode.NONE, {})):
                d = ImageDitherer(K, mode, pal, False, params).prepare()
                res = []
                for i, env in enumerate(({}, {"DP_NO_COMPACT_KERNEL": "1"}, {THIRD: "1"})):
                    for k, v in env.items(): os.environ[k] = v
                    res.append(timeit(lambda: d.apply_dithering_frames(f, out=outs[i])))
                    for k in env: del os.environ[k]
                same = torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
                print(f"{kind:7s} K={K:3d} {src:10s} {mode.value:6s}: compact {res[0]:6.3f} ms ({24*2160*3840/res[0]/1e6:6.1f} Gpx/s) | lean adaptive {res[1]:6.3f} ms | "
                      f"compact, {THIRD}=1 {res[2]:6.3f} ms | identical bytes: {same}", flush=True)
