"""Variable-weight diffusers on image-like frames vs noise (how often a wave has a point outside the colour cube)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
pal = ColorReducer.generate_uniform_palette(16)
rs = np.random.RandomState(3)
y, x = np.mgrid[0:540, 0:960]
a = np.clip(np.stack([80 + 60 * np.sin(x / 300.0) + 40 * (y / 540), 110 + 50 * np.cos(y / 200.0) + 20 * np.sin(x / 97.0),
                      160 + 70 * (y / 540) + 10 * np.sin((x + y) / 50.0)], -1) + rs.normal(0, 3, (540, 960, 3)), 0, 255).astype(np.uint8)
img = torch.from_numpy(a).cuda().repeat(4, 4, 1).unsqueeze(0).repeat(64, 1, 1, 1).contiguous()
g = torch.Generator(device='cuda'); g.manual_seed(1)
noise = torch.randint(0, 256, (64, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g)
o = torch.empty_like(noise)
for mode in (DitherMode.PERCEPTUAL, DitherMode.HYBRID, DitherMode.ADAPTIVE_VARIANCE, DitherMode.OSTROMOUKHOV):
    d = ImageDitherer(16, mode, pal, False, {})
    res = []
    for f in (img, noise):
        d.apply_dithering_frames(f, out=o); torch.cuda.synchronize()
        t0 = time.perf_counter(); d.apply_dithering_frames(f, out=o); torch.cuda.synchronize(); res.append(time.perf_counter() - t0)
    print(f"{mode.value:18s} 64 4K frames: image-like {res[0]*1e3:7.1f} ms, noise {res[1]*1e3:7.1f} ms", flush=True)
