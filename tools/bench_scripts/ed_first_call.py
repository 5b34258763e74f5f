"""What the FIRST error-diffusion call on a fresh palette pays on top of the frame (KD-tree, 8^3 cell lists and their octree, 16^3
lists, hierarchical table -- host_logic.h: ed_tables_refine; built on first use): first against second call, one 4K frame."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from PIL import Image
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
g = torch.Generator(device='cuda'); g.manual_seed(1)
f = torch.randint(0, 256, (1, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
warm = ImageDitherer(16, DitherMode.ERROR_DIFFUSION, ColorReducer.generate_uniform_palette(16), False, {"variant": "floyd_steinberg", "serpentine": "false"})
warm.apply_dithering_frames(f, out=o); torch.cuda.synchronize()
yy, xx = np.mgrid[0:540, 0:960]
img = np.clip(np.stack([80 + 60 * np.sin(xx / 300.0) + 40 * (yy / 540.0), 110 + 50 * np.cos(yy / 200.0), 160 + 70 * (yy / 540.0)], -1) + np.random.RandomState(3).normal(0, 3, (540, 960, 3)), 0, 255).astype(np.uint8)
cases = [(f"random {K}", [tuple(int(v) for v in c) for c in np.random.RandomState(20 + K).randint(0, 256, (K, 3))]) for K in (16, 32, 64, 128, 256, 1024)]
cases += [(f"median cut {K}", ColorReducer.reduce_colors(Image.fromarray(img, "RGB"), K)) for K in (16, 64, 256)]
for name, pal in cases:
    d = ImageDitherer(len(pal), DitherMode.ERROR_DIFFUSION, pal, False, {"variant": "floyd_steinberg", "serpentine": "false"})
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); d.apply_dithering_frames(f, out=o); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(f"{name:16s} first call {ts[0]:8.2f} ms   second {ts[1]:7.2f}   third {ts[2]:7.2f}   -> tables {ts[0] - ts[2]:7.2f} ms", flush=True)
