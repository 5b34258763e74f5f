import sys; sys.path.insert(0,'.')
import numpy as np, torch
from PIL import Image
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
rs=np.random.RandomState(3)
h,w=540,960
y,x=np.mgrid[0:h,0:w]
r=80+60*np.sin(x/300.0)+40*(y/h); g=110+50*np.cos(y/200.0)+20*np.sin(x/97.0); b=160+70*(y/h)+10*np.sin((x+y)/50.0)
a=np.clip(np.stack([r,g,b],-1)+rs.normal(0,3,(h,w,3)),0,255).astype(np.uint8)
pal=ColorReducer.reduce_colors(Image.fromarray(a,"RGB"),256)
big=torch.from_numpy(a).cuda().repeat(4,4,1).unsqueeze(0).repeat(24,1,1,1).contiguous(); out=torch.empty_like(big)
d=ImageDitherer(256,DitherMode.BAYER,pal,False,{"size":"8x8"})
for _ in range(6): d.apply_dithering_frames(big,out=out)
torch.cuda.synchronize()
