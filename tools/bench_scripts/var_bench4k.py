import sys, time; sys.path.insert(0,'.')
import torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
pal=ColorReducer.generate_uniform_palette(16)
g=torch.Generator(device='cuda'); g.manual_seed(1)
f=torch.randint(0,256,(256,2160,3840,3),dtype=torch.uint8,device='cuda',generator=g); o=torch.empty_like(f)
for mode,params in [(DitherMode.PERCEPTUAL,{}),(DitherMode.HYBRID,{}),(DitherMode.ADAPTIVE_VARIANCE,{}),(DitherMode.OSTROMOUKHOV,{})]:
    d=ImageDitherer(16, mode, pal, False, params)
    d.apply_dithering_frames(f[:2],out=o[:2]); torch.cuda.synchronize()
    t0=time.perf_counter(); d.apply_dithering_frames(f[:1],out=o[:1]); torch.cuda.synchronize(); t1=time.perf_counter()-t0
    d.apply_dithering_frames(f,out=o); torch.cuda.synchronize()  # first pass at this size allocates the workspaces (17 GB for the variance gate)
    t0=time.perf_counter(); d.apply_dithering_frames(f,out=o); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(f"{mode.value:18s} 4K: single frame {t1*1e3:7.1f} ms; 256 frames {dt*1e3:8.1f} ms = {256*2160*3840/dt/1e9:6.2f} Gpx/s", flush=True)
