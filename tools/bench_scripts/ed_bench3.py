import sys, time; sys.path.insert(0,'.')
import torch, numpy as np
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
g=torch.Generator(device='cuda'); g.manual_seed(1)
f=torch.randint(0,256,(256,2160,3840,3),dtype=torch.uint8,device='cuda',generator=g); o=torch.empty_like(f)
for K,variant in [(16,"floyd_steinberg"),(16,"jjn"),(16,"atkinson"),(256,"floyd_steinberg"),(2,"floyd_steinberg")]:
    pal=ColorReducer.generate_uniform_palette(K) if K<=64 else [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0,256,(K,3))]
    d=ImageDitherer(K, DitherMode.ERROR_DIFFUSION, pal, False, {"variant":variant,"serpentine":"false"})
    d.apply_dithering_frames(f[:2],out=o[:2]); torch.cuda.synchronize()
    t0=time.perf_counter(); d.apply_dithering_frames(f[:1],out=o[:1]); torch.cuda.synchronize(); t1=time.perf_counter()-t0
    nf=256 if K<=16 else 64
    t0=time.perf_counter(); d.apply_dithering_frames(f[:nf],out=o[:nf]); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(f"{variant:16s} K={K:3d} 4K: single {t1*1e3:7.1f} ms; {nf} frames {dt*1e3:8.1f} ms = {nf*2160*3840/dt/1e9:6.2f} Gpx/s", flush=True)
