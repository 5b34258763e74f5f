"""Kernel time of the use_gamma (float palette) ordered kernel and the image-derived-palette cases on the C2 batch."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import backend as be
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
pal = [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (256, 3))]
g = torch.Generator(device='cuda'); g.manual_seed(1)
f = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
for name, mode, params, gamma in [("bayer8 gamma", DitherMode.BAYER, {"size": "8x8"}, True), ("none gamma", DitherMode.NONE, {}, True),
                                  ("ign gamma", DitherMode.INTERLEAVED_GRADIENT_NOISE, {}, True), ("blue gamma", DitherMode.BLUE_NOISE, {"size": 64}, True)]:
    d = ImageDitherer(256, mode, pal, gamma, params).prepare()
    for _ in range(3): d.apply_dithering_frames(f, out=o)
    ms, fx = [], []
    for _ in range(7):
        be.profile_enable(True); d.apply_dithering_frames(f, out=o); torch.cuda.synchronize()
        m, x, n = be.profile_read(); be.profile_enable(False); ms.append(m / max(n, 1)); fx.append(x / max(n, 1))
    ms.sort(); fx.sort()
    print(f"{name:14s} main {ms[len(ms)//2]:.4f} ms  fix-up {fx[len(fx)//2]:.4f} ms   {f.numel()/3/(ms[len(ms)//2]+fx[len(fx)//2])/1e6:.1f} Gpx/s", flush=True)
