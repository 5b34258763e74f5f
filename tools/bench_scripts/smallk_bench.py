import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
g=torch.Generator(device='cuda'); g.manual_seed(1)
F=24
f=torch.randint(0,256,(F,2160,3840,3),dtype=torch.uint8,device='cuda',generator=g); o=torch.empty_like(f)
for K in (2,4,7,8,16,64,300,1024):
    pal=[tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0,256,(K,3))]
    for name,mode,params in (("bayer8",DitherMode.BAYER,{"size":"8x8"}),("none",DitherMode.NONE,{})):
        d=ImageDitherer(K, mode, pal, False, params)
        d.apply_dithering_frames(f,out=o); torch.cuda.synchronize()
        n=3
        t0=time.perf_counter()
        for _ in range(n): d.apply_dithering_frames(f,out=o)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/n
        print(f"K={K:4d} {name:7s}: {dt*1e3:8.3f} ms / {F} frames = {f.numel()/3/dt/1e9:7.2f} Gpx/s", flush=True)
