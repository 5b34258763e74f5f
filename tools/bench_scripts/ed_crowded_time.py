"""Error diffusion with palettes extracted from the content (median cut -- the reference's default palette source) on image-like 4K
frames, against uniform / random palettes of the same size on the same frames: one frame, 24 frames.  usage: ed_crowded_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["DITHER_PIE_EXPERIMENTS"] = "1"; os.environ["DP_ED_H4_REPORT"] = "1"
import numpy as np, torch
from PIL import Image
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer

rs = np.random.RandomState(3)
yy, xx = np.mgrid[0:540, 0:960]
img = np.clip(np.stack([80 + 60 * np.sin(xx / 300.0) + 40 * (yy / 540.0), 110 + 50 * np.cos(yy / 200.0) + 20 * np.sin(xx / 97.0),
                        160 + 70 * (yy / 540.0) + 10 * np.sin((xx + yy) / 50.0)], -1) + rs.normal(0, 3, (540, 960, 3)), 0, 255).astype(np.uint8)
frames = torch.from_numpy(img).cuda().repeat(4, 4, 1).unsqueeze(0).repeat(24, 1, 1, 1).contiguous()
out = torch.empty_like(frames)


def t(d, n, reps=3):
    d.apply_dithering_frames(frames[:n], out=out[:n]); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); d.apply_dithering_frames(frames[:n], out=out[:n]); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for K in (16, 64, 256):
    pals = {"median cut": ColorReducer.reduce_colors(Image.fromarray(img, "RGB"), K),
            "uniform" if K <= 64 else "random": (ColorReducer.generate_uniform_palette(K) if K <= 64 else
                                                 [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (K, 3))])}
    for name, pal in pals.items():
        d = ImageDitherer(K, DitherMode.ERROR_DIFFUSION, pal, False, {"variant": "floyd_steinberg", "serpentine": "false"})
        print(f"K={K:3d} {name:10s}: one 4K frame {t(d, 1):7.2f} ms   24 frames {t(d, 24):7.2f} ms", flush=True)
