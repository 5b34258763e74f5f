"""C2-shaped batch of image-like frames with their own 256-colour median-cut palette (the crowded-palette case of bench.py):
a few launches for rocprofv3, plus event timing.  usage: crowded_prof.py [launches]"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from PIL import Image
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rs = np.random.RandomState(3)
yy, xx = np.mgrid[0:540, 0:960]
img = np.clip(np.stack([80 + 60 * np.sin(xx / 300.0) + 40 * (yy / 540.0), 110 + 50 * np.cos(yy / 200.0) + 20 * np.sin(xx / 97.0),
                        160 + 70 * (yy / 540.0) + 10 * np.sin((xx + yy) / 50.0)], -1) + rs.normal(0, 3, (540, 960, 3)), 0, 255).astype(np.uint8)
pal = ColorReducer.reduce_colors(Image.fromarray(img, "RGB"), 256)
f = torch.from_numpy(img).cuda().repeat(4, 4, 1).unsqueeze(0).repeat(24, 1, 1, 1).contiguous(); o = torch.empty_like(f)
d = ImageDitherer(256, DitherMode.BAYER, pal, False, {"size": "8x8"}).prepare()
for _ in range(3): d.apply_dithering_frames(f, out=o)
ts = []
for _ in range(n):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); d.apply_dithering_frames(f, out=o); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(f"crowded median-cut 256, bayer8, 24 x 4K: min {min(ts):.3f} ms = {24*2160*3840/min(ts)/1e6:.1f} Gpx/s", flush=True)
