"""One ordered mode, C2-shaped (24 x 4K noise frames, 256 random colours), a few launches: the target of rocprofv3 passes.
usage: python3 tools/bench_scripts/prof_ordered.py <none|bayer8|bayer4|blue|ign> [launches]"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
which = sys.argv[1] if len(sys.argv) > 1 else "bayer8"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pal = [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (256, 3))]
g = torch.Generator(device='cuda'); g.manual_seed(1)
f = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
mode, params = {"none": (DitherMode.NONE, {}), "bayer8": (DitherMode.BAYER, {"size": "8x8"}), "bayer4": (DitherMode.BAYER, {"size": "4x4"}),
                "blue": (DitherMode.BLUE_NOISE, {"size": 64}), "ign": (DitherMode.INTERLEAVED_GRADIENT_NOISE, {})}[which]
d = ImageDitherer(256, mode, pal, False, params)
for _ in range(n): d.apply_dithering_frames(f, out=o)
torch.cuda.synchronize()
print("done", which, n)
