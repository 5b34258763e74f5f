"""Kernel time (library HIP events) of the ordered modes on the C2 batch: 24 x 4K noise frames, 256 random colours."""
import os, sys; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import backend as be
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
pal = [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (256, 3))]
g = torch.Generator(device='cuda'); g.manual_seed(1)
f = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
allc = {"none": (DitherMode.NONE, {}), "bayer8": (DitherMode.BAYER, {"size": "8x8"}), "bayer4": (DitherMode.BAYER, {"size": "4x4"}),
        "bayer16": (DitherMode.BAYER, {"size": "16x16"}), "bayer2": (DitherMode.BAYER, {"size": "2x2"}),
        "blue": (DitherMode.BLUE_NOISE, {"size": 64}), "ign": (DitherMode.INTERLEAVED_GRADIENT_NOISE, {}), "polka": (DitherMode.POLKA_DOT, {})}
for name in (sys.argv[1:] or list(allc)):
    mode, params = allc[name]
    d = ImageDitherer(256, mode, pal, False, params)
    for _ in range(10): d.apply_dithering_frames(f, out=o)
    ms = []
    for _ in range(15):
        be.profile_enable(True); d.apply_dithering_frames(f, out=o); torch.cuda.synchronize()
        m, fx, n = be.profile_read(); be.profile_enable(False); ms.append(m / max(n, 1))
    ms.sort()
    print(f"{name:8s} min {ms[0]:.4f}  med {ms[len(ms)//2]:.4f} ms   {f.numel()/3/ms[len(ms)//2]/1e6:.1f} Gpx/s", flush=True)
