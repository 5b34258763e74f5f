"""C2 with use_gamma=True (float palette, ordered_compact_float_kernel): a few launches for rocprofv3, plus event timing.
usage: gamma_prof.py [launches]"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pal = [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (256, 3))]
g = torch.Generator(device='cuda'); g.manual_seed(1234)
f = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
d = ImageDitherer(256, DitherMode.BAYER, pal, True, {"size": "8x8"}).prepare()
for _ in range(3): d.apply_dithering_frames(f, out=o)
ts = []
for _ in range(n):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); d.apply_dithering_frames(f, out=o); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(f"use_gamma, palr(256), bayer8, 24 x 4K: min {min(ts):.3f} ms = {24*2160*3840/min(ts)/1e6:.1f} Gpx/s", flush=True)
