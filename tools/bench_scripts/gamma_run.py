import sys; sys.path.insert(0,'.')
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
pal=[tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0,256,(256,3))]
g=torch.Generator(device='cuda'); g.manual_seed(1234)
f=torch.randint(0,256,(24,2160,3840,3),dtype=torch.uint8,device='cuda',generator=g); o=torch.empty_like(f)
d=ImageDitherer(256, DitherMode.BAYER, pal, True, {"size":"8x8"})
for _ in range(6): d.apply_dithering_frames(f,out=o)
torch.cuda.synchronize()
