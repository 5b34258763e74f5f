"""64 4K frames through one variable-coefficient diffuser (for rocprofv3 --pmc).  usage: var_prof.py [mode]"""
import sys; sys.path.insert(0, '.')
import torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
mode = DitherMode(sys.argv[1]) if len(sys.argv) > 1 else DitherMode.PERCEPTUAL
pal = ColorReducer.generate_uniform_palette(16)
g = torch.Generator(device='cuda'); g.manual_seed(1)
d = ImageDitherer(16, mode, pal, False, {})
f = torch.randint(0, 256, (64, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
d.apply_dithering_frames(f, out=o); torch.cuda.synchronize()
d.apply_dithering_frames(f, out=o); torch.cuda.synchronize()
