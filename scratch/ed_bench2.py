import sys, time; sys.path.insert(0,'.')
import torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
pal=ColorReducer.generate_uniform_palette(16)
g=torch.Generator(device='cuda'); g.manual_seed(1)
for variant in ("floyd_steinberg","jjn","atkinson"):
    d=ImageDitherer(16, DitherMode.ERROR_DIFFUSION, pal, False, {"variant":variant,"serpentine":"false"})
    for n in (128,256,512):
        f=torch.randint(0,256,(n,2160,3840,3),dtype=torch.uint8,device='cuda',generator=g); o=torch.empty_like(f)
        d.apply_dithering_frames(f,out=o); torch.cuda.synchronize()
        t0=time.perf_counter(); d.apply_dithering_frames(f,out=o); torch.cuda.synchronize(); dt=time.perf_counter()-t0
        print(f"{variant:16s} {n} frames 4K: {dt*1e3:8.1f} ms  {n*2160*3840/dt/1e9:6.2f} Gpx/s", flush=True)
        del f,o
