import sys, time; sys.path.insert(0,'.')
import torch, numpy as np
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
from dither_pie_amd import backend
pal=ColorReducer.generate_uniform_palette(16)
d=ImageDitherer(16, DitherMode.ERROR_DIFFUSION, pal, False, {"variant":"floyd_steinberg","serpentine":"false"})
g=torch.Generator(device='cuda'); g.manual_seed(1)
for (n,h,w) in [(1,2160,3840),(8,2160,3840),(64,2160,3840),(1,1080,1920),(1,256,256)]:
    f=torch.randint(0,256,(n,h,w,3),dtype=torch.uint8,device='cuda',generator=g); o=torch.empty_like(f)
    d.apply_dithering_frames(f,out=o); torch.cuda.synchronize()
    t0=time.perf_counter(); d.apply_dithering_frames(f,out=o); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(f"FS K=16 {n}x{h}x{w}: {dt*1e3:9.2f} ms  {n*h*w/dt/1e6:9.1f} Mpx/s", flush=True)
