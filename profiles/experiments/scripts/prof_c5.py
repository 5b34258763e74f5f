"""C5-shaped launches for the counter passes (profiles/pmc_pass_cmd4.sh): 100 frames of 1920x1080 noise, Bayer 4x4 (or the mode
given), 16 uniform colours -- ordered_lean_kernel<1,4,HALF>.   usage: prof_c5.py [launches=6] [K=16] [bayer4|bayer8|none|ign]"""
import sys; sys.path.insert(0, '.')
import torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
which = sys.argv[3] if len(sys.argv) > 3 else "bayer4"
mode, params = {"none": (DitherMode.NONE, {}), "bayer8": (DitherMode.BAYER, {"size": "8x8"}), "bayer4": (DitherMode.BAYER, {"size": "4x4"}),
                "ign": (DitherMode.INTERLEAVED_GRADIENT_NOISE, {})}[which]
g = torch.Generator(device='cuda'); g.manual_seed(1)
f = torch.randint(0, 256, (100, 1080, 1920, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
d = ImageDitherer(K, mode, ColorReducer.generate_uniform_palette(K), False, params).prepare()  # (the cell table at once, as a video has it)
for _ in range(n): d.apply_dithering_frames(f, out=o)
torch.cuda.synchronize()
print("done", n, K, which)
