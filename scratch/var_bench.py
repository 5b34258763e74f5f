import sys, time; sys.path.insert(0,'.')
import torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
pal=ColorReducer.generate_uniform_palette(16)
g=torch.Generator(device='cuda'); g.manual_seed(1)
f=torch.randint(0,256,(1024,270,480,3),dtype=torch.uint8,device='cuda',generator=g); o=torch.empty_like(f)
for mode,params in [(DitherMode.PERCEPTUAL,{}),(DitherMode.HYBRID,{}),(DitherMode.ADAPTIVE_VARIANCE,{}),(DitherMode.OSTROMOUKHOV,{}),(DitherMode.ERROR_DIFFUSION,{"variant":"floyd_steinberg","serpentine":"true"})]:
    d=ImageDitherer(16, mode, pal, False, params)
    d.apply_dithering_frames(f[:64],out=o[:64]); torch.cuda.synchronize()
    t0=time.perf_counter(); d.apply_dithering_frames(f,out=o); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(f"{mode.value:18s} 1024 frames 480x270: {dt*1e3:8.1f} ms  {1024*270*480/dt/1e6:8.1f} Mpx/s", flush=True)
