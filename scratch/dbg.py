import sys; sys.path.insert(0,'.')
import numpy as np, torch
from PIL import Image
from oracle import oracle as orc
from dither_pie_amd import dithering_lib as d, video_processor as v, backend
pal = orc.generate_uniform_palette(16)
it = d.ImageDitherer(16, d.DitherMode.BAYER, pal, False, {"size": "4x4"})
frames = np.stack([orc.rnd(60, 80, s) for s in range(3)])
x=torch.from_numpy(frames).cuda()
tw,th=v._even_dimensions(80,60,16)
small=backend.resize_nearest(x, th, tw)
for i in range(3):
    ref=np.array(Image.fromarray(frames[i]).resize((tw,th), Image.NEAREST))
    print('resize', i, np.array_equal(small[i].cpu().numpy(), ref))
dith=it.apply_dithering_frames(small).cpu().numpy()
for i in range(3):
    ref=orc.apply_dithering(small[i].cpu().numpy(), pal, "bayer", {"size":"4x4"})
    bad=np.argwhere((dith[i]!=ref).any(-1)); print('dither', i, len(bad), bad[:5].tolist())
    one=it.apply_dithering_frames(small[i].contiguous()).cpu().numpy()
    print('  single-frame equal to ref:', np.array_equal(one, ref))
big=backend.resize_nearest(torch.from_numpy(dith).cuda(), 32, 44).cpu().numpy()
for i in range(3):
    ref=np.array(Image.fromarray(dith[i]).resize((44,32), Image.NEAREST)); print('up', i, np.array_equal(big[i], ref))
