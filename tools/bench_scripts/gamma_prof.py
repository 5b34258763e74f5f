import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from dither_pie_amd import backend
from dither_pie_amd.dithering_lib import prepare_palette, DitherUtils
pal=[tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0,256,(256,3))]
g=torch.Generator(device='cuda'); g.manual_seed(1234)
f=torch.randint(0,256,(24,2160,3840,3),dtype=torch.uint8,device='cuda',generator=g)
npx=24*2160*3840
for gamma in (True, False):
    P=backend.Palette(*prepare_palette(pal,gamma), accel=True)
    print("gamma",gamma,"accel entries",P.accel_entries,"max list",P.accel_max_list)
    thr=backend.Thresholds.from_matrix(DitherUtils.BAYER8x8)
    for mode,name in ((backend.MODE_MATRIX,"bayer8"),(backend.MODE_NEAREST,"none")):
        out=backend.ordered(f,P,mode,thr=thr if mode==backend.MODE_MATRIX else None); torch.cuda.synchronize()
        ws=list(backend._ws_cache.values())[0]
        dirty_off=(((npx+255)//256*32+768)&~7)
        d=ws[dirty_off:dirty_off+8].view(torch.int32)
        backend.profile_enable(True); backend.profile_read()
        for _ in range(5): backend.ordered(f,P,mode,thr=thr if mode==backend.MODE_MATRIX else None)
        torch.cuda.synchronize(); a,b,n=backend.profile_read(); backend.profile_enable(False)
        print(f"  {name}: main {a/n:.3f} ms fixup {b/n:.3f} ms; dirty wave tiles {int(d[0])}")
