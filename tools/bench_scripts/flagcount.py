import sys; sys.path.insert(0,'.')
import torch, numpy as np
from dither_pie_amd import backend
from dither_pie_amd.dithering_lib import prepare_palette, DitherUtils
pal=[tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0,256,(256,3))]
P=backend.Palette(*prepare_palette(pal,False), accel=True)
thr=backend.Thresholds.from_matrix(DitherUtils.BAYER8x8)
g=torch.Generator(device='cuda'); g.manual_seed(1234)
f=torch.randint(0,256,(24,2160,3840,3),dtype=torch.uint8,device='cuda',generator=g)
out=backend.ordered(f,P,backend.MODE_MATRIX,thr=thr); torch.cuda.synchronize()
ws=list(backend._ws_cache.values())[0]
npx=24*2160*3840
words=ws[: (npx+4095)//4096*4096//8].view(torch.int64)
bits=sum(int(((words>>k)&1).sum()) for k in range(64))
print("flagged pixels:", bits, "of", npx, f"= {bits/npx*100:.4f}%")
P2=backend.Palette(*prepare_palette(pal,False), accel=False)
out2=backend.ordered(f,P2,backend.MODE_MATRIX,thr=thr); torch.cuda.synchronize()
print("accel == brute force:", bool(torch.equal(out,out2)))
dirty_off=(( (npx+255)//256*32 + 768) & ~7)
d=ws[dirty_off:dirty_off+8].view(torch.int32)
print("dirty count (queued wave tiles):", int(d[0]))
