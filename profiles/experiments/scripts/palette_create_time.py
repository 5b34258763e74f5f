"""Time of dp_palette_create (KD-tree + the error-diffusion candidate tables) for a fresh palette of K colours."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import backend as be
torch.cuda.init(); torch.zeros(1).cuda()
for K in (2, 8, 16, 64, 256, 1024):
    ts = []
    for rep in range(5):
        pal = np.random.RandomState(100 + rep + K).randint(0, 256, (K, 3)).astype(np.float32)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        P = be.Palette(pal, pal.astype(np.uint8), None)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        if rep == 4:
            from dither_pie_amd.dithering_lib import ErrorDiffusionKernel
            k = ErrorDiffusionKernel.get_kernel("floyd_steinberg")
            f = torch.zeros((1, 64, 64, 3), dtype=torch.uint8, device="cuda")
            be.error_diffusion(f, P, k["weights"], k["divisor"]); torch.cuda.synchronize()  # (library load etc.)
            P2 = be.Palette(pal + 0, pal.astype(np.uint8), None)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            be.error_diffusion(f, P2, k["weights"], k["divisor"]); torch.cuda.synchronize()
            first_ed = (time.perf_counter() - t0) * 1e3
        del P
    print(f"K={K:5d}: dp_palette_create min {min(ts):7.2f} ms  median {sorted(ts)[2]:7.2f} ms;  first diffusion call with a new palette (builds its candidate tables) {first_ed:7.2f} ms", flush=True)
