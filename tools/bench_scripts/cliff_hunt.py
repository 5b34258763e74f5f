"""One 1080p frame through every mode x palette size x use_gamma (second and third call: tables exist): a search for cliffs --
anything far from its neighbours in the table.  usage: cliff_hunt.py [frames]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = torch.Generator(device='cuda'); g.manual_seed(1)
f = torch.randint(0, 256, (n, 1080, 1920, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
modes = [("none", {}), ("bayer", {"size": "8x8"}), ("blue_noise", {"size": 64, "seed": 42}), ("IGN", {}), ("polka_dot", {}),
         ("error_diffusion", {"variant": "floyd_steinberg", "serpentine": "false"}), ("error_diffusion", {"variant": "jjn", "serpentine": "false"}),
         ("error_diffusion", {"variant": "atkinson", "serpentine": "true"}), ("perceptual", {}), ("hybrid", {}), ("adaptive_variance", {}),
         ("ostromoukhov", {"serpentine": "false"}), ("ostromoukhov", {"serpentine": "true"})]
Ks = (2, 3, 4, 8, 9, 16, 17, 64, 256, 257, 1024)
print(f"{n} x 1080p, ms per call (3rd call); rows: mode, columns: K = " + " ".join(str(k) for k in Ks))
for gamma in (False, True):
    for mode, params in modes:
        row = []
        for K in Ks:
            pal = [tuple(int(v) for v in c) for c in np.random.RandomState(100 + K).randint(0, 256, (K, 3))]
            try:
                d = ImageDitherer(K, DitherMode(mode), pal, gamma, dict(params))
                for _ in range(2): d.apply_dithering_frames(f, out=o)
                torch.cuda.synchronize(); t = time.perf_counter(); d.apply_dithering_frames(f, out=o); torch.cuda.synchronize()
                row.append((time.perf_counter() - t) * 1e3)
            except Exception as e:  # noqa: BLE001
                row.append(float("nan")); print("  !!", mode, K, gamma, type(e).__name__, str(e)[:80])
        tag = f"{mode}{'/serp' if params.get('serpentine') == 'true' else ''}{'/' + params['variant'] if 'variant' in params else ''}{' gamma' if gamma else ''}"
        print(f"{tag:42s}" + " ".join(f"{v:8.2f}" for v in row), flush=True)
