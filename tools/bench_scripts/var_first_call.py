import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
g = torch.Generator(device='cuda'); g.manual_seed(1)
f = torch.randint(0, 256, (1, 1080, 1920, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
ImageDitherer(16, DitherMode("perceptual"), [tuple(int(v) for v in c) for c in np.random.RandomState(1).randint(0, 256, (16, 3))], False, {}).apply_dithering_frames(f, out=o); torch.cuda.synchronize()
for K in (64, 256, 1024):
    for gamma in (False, True):
        pal = [tuple(int(v) for v in c) for c in np.random.RandomState(50 + K).randint(0, 256, (K, 3))]
        d = ImageDitherer(K, DitherMode("perceptual"), pal, gamma, {})
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter(); d.apply_dithering_frames(f, out=o); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        print(f"perceptual K={K} gamma={gamma}: first {ts[0]:.1f} ms, third {ts[2]:.1f} -> tables {ts[0]-ts[2]:.1f} ms", flush=True)
