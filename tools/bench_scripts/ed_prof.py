"""A few launches of Floyd-Steinberg at 4K for the counter passes (profiles/pmc_pass_cmd.sh, pmc_pass_cmd4.sh).
usage: ed_prof.py [K=16] [frames=64]     (K <= 64: the uniform palette, else palr(K, 7); frames = 1: the few-frames schedule)"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 64
pal = ColorReducer.generate_uniform_palette(K) if K <= 64 else [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (K, 3))]
g = torch.Generator(device='cuda'); g.manual_seed(1)
d = ImageDitherer(K, DitherMode.ERROR_DIFFUSION, pal, False, {"variant": "floyd_steinberg", "serpentine": "false"})
f = torch.randint(0, 256, (nf, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
d.apply_dithering_frames(f, out=o); torch.cuda.synchronize()
d.apply_dithering_frames(f, out=o); torch.cuda.synchronize()
