"""The headline batch (24 4K frames rnd, palr(256), Bayer 8x8 / nearest / IGN) on the lean kernels and, forced, on the compact
kernel (DP_FORCE_COMPACT=1 at palette build and launch).  usage: compact_headline.py"""
import os, sys; sys.path.insert(0, '.')
os.environ["DITHER_PIE_EXPERIMENTS"] = "1"
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
def palr(K, seed=7): return [tuple(int(v) for v in c) for c in np.random.RandomState(seed).randint(0, 256, (K, 3))]
g = torch.Generator(device='cuda'); g.manual_seed(1234)
f = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g)
def timeit(fn, n=8):
    for _ in range(3): fn()
    ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
for K in (256, 64, 16):
    for mode, params in ((DitherMode.BAYER, {"size": "8x8"}), (DitherMode.NONE, {}), (DitherMode.INTERLEAVED_GRADIENT_NOISE, {})):
        outs = []; res = []
        for force in (False, True):
            from dither_pie_amd import dithering_lib as dl
            dl._PALETTES.clear()
            if force: os.environ["DP_FORCE_COMPACT"] = "1"
            d = ImageDitherer(K, mode, palr(K), False, params).prepare()
            o = torch.empty_like(f)
            res.append(timeit(lambda: d.apply_dithering_frames(f, out=o))); outs.append(o)
            os.environ.pop("DP_FORCE_COMPACT", None)
            from dither_pie_amd import dithering_lib as dl
            dl._PALETTES.clear()   # the forced build must not reuse the palette object built without the switch
        print(f"K={K:3d} {mode.value:6s}: lean {res[0]:.3f} ms | compact (forced) {res[1]:.3f} ms | identical: {torch.equal(outs[0], outs[1])}", flush=True)
