"""Does the cell table's LDS bank mapping matter?  Times the headline kernel on frames whose pixels fall into chosen sets
of cells: one cell (broadcast reads), cells that differ only in g' / b' (same banks), only in r' (spread over banks)."""
import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
rs=np.random.RandomState(7)
pal=[tuple(int(v) for v in c) for c in rs.randint(0,256,(256,3))]
d=ImageDitherer(256,DitherMode.BAYER,pal,False,{"size":"8x8"})
g=torch.Generator(device='cuda'); g.manual_seed(1)
def frames(rvar,gvar,bvar):
    n=(24,2160,3840)
    lo=torch.randint(0,16,n+(3,),dtype=torch.uint8,device='cuda',generator=g)
    hi=torch.randint(0,16,n+(3,),dtype=torch.uint8,device='cuda',generator=g)
    keep=torch.tensor([rvar,gvar,bvar],dtype=torch.uint8,device='cuda')
    return (lo+((hi*keep+8*(1-keep))<<4)).contiguous()
out=None
for name,v in (("uniform noise",(1,1,1)),("one cell",(0,0,0)),("g' varies",(0,1,0)),("b' varies",(0,0,1)),("r' varies",(1,0,0)),("g',b' vary",(0,1,1))):
    f=frames(*v)
    if out is None: out=torch.empty_like(f)
    for _ in range(5): d.apply_dithering_frames(f,out=out)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): d.apply_dithering_frames(f,out=out)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print(f"{name:14s}: {dt*1e3:.3f} ms", flush=True)
