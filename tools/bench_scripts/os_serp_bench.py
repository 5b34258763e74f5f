import sys, time; sys.path.insert(0,'.')
import torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
g=torch.Generator(device='cuda'); g.manual_seed(1)
pal=ColorReducer.generate_uniform_palette(16)
d=ImageDitherer(16, DitherMode.OSTROMOUKHOV, pal, False, {"serpentine":"true"})
for n,h,w in [(1,270,480),(1,1080,1920),(64,1080,1920)]:
    f=torch.randint(0,256,(n,h,w,3),dtype=torch.uint8,device='cuda',generator=g); o=torch.empty_like(f)
    d.apply_dithering_frames(f[:1,:32],out=o[:1,:32]); torch.cuda.synchronize()
    t0=time.perf_counter(); d.apply_dithering_frames(f,out=o); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(f"ostromoukhov serpentine K=16 {n}x{h}x{w}: {dt*1e3:9.1f} ms  {n*h*w/dt/1e6:9.1f} Mpx/s  ({dt/(h*w)*1e9:.0f} ns per pixel step)", flush=True)
