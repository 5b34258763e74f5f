import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
pal=[tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0,256,(256,3))]
g=torch.Generator(device='cuda'); g.manual_seed(1)
F=24
f=torch.randint(0,256,(F,2160,3840,3),dtype=torch.uint8,device='cuda',generator=g); o=torch.empty_like(f)
f2=f[:,:,:3838].contiguous(); o2=torch.empty_like(f2)
cases=[("none",DitherMode.NONE,{},False,f,o),("bayer8",DitherMode.BAYER,{"size":"8x8"},False,f,o),
       ("bayer8 w=3838",DitherMode.BAYER,{"size":"8x8"},False,f2,o2),
       ("blue_noise64",DitherMode.BLUE_NOISE,{"size":64},False,f,o),("IGN",DitherMode.INTERLEAVED_GRADIENT_NOISE,{},False,f,o),
       ("polka",DitherMode.POLKA_DOT,{},False,f,o),
       ("bayer8 gamma",DitherMode.BAYER,{"size":"8x8"},True,f,o),("none gamma",DitherMode.NONE,{},True,f,o)]
for name,mode,params,gamma,fi,oo in cases:
    try:
        d=ImageDitherer(256, mode, pal, gamma, params)
        d.apply_dithering_frames(fi[:2],out=oo[:2]); torch.cuda.synchronize()
        d.apply_dithering_frames(fi,out=oo); torch.cuda.synchronize()
        n=5 if not gamma else 2
        t0=time.perf_counter()
        for _ in range(n): d.apply_dithering_frames(fi,out=oo)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/n
        print(f"{name:16s}: {dt*1e3:8.3f} ms / {F} frames = {fi.numel()/3/dt/1e9:7.2f} Gpx/s", flush=True)
    except Exception as e:
        print(name, "ERR", repr(e)[:200], flush=True)
