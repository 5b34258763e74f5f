"""use_gamma and error diffusion with palettes extracted from image-like content."""
import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from PIL import Image
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
rs=np.random.RandomState(3)
h,w=540,960
y,x=np.mgrid[0:h,0:w]
r=80+60*np.sin(x/300.0)+40*(y/h); g=110+50*np.cos(y/200.0)+20*np.sin(x/97.0); b=160+70*(y/h)+10*np.sin((x+y)/50.0)
a=np.clip(np.stack([r,g,b],-1)+rs.normal(0,3,(h,w,3)),0,255).astype(np.uint8)
big=torch.from_numpy(a).cuda().repeat(4,4,1).unsqueeze(0).repeat(24,1,1,1).contiguous(); out=torch.empty_like(big)
for K in (16,256):
    pal=ColorReducer.reduce_colors(Image.fromarray(a,"RGB"),K)
    for name,mode,params,gamma,nf in (("bayer gamma",DitherMode.BAYER,{"size":"8x8"},True,24),("FS",DitherMode.ERROR_DIFFUSION,{"variant":"floyd_steinberg","serpentine":"false"},False,24),
                                      ("FS gamma",DitherMode.ERROR_DIFFUSION,{"variant":"floyd_steinberg","serpentine":"false"},True,24)):
        d=ImageDitherer(K,mode,pal,gamma,params)
        d.apply_dithering_frames(big[:nf],out=out[:nf]); torch.cuda.synchronize()
        t0=time.perf_counter(); d.apply_dithering_frames(big[:nf],out=out[:nf]); torch.cuda.synchronize(); dt=time.perf_counter()-t0
        print(f"smooth content, median-cut K={K:3d}, {name:12s}: {dt*1e3:8.2f} ms / {nf} 4K frames = {nf*2160*3840/dt/1e9:7.2f} Gpx/s", flush=True)
