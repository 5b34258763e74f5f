"""Cell-table statistics and timing for palettes extracted from image-like content (clustered colours)."""
import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from PIL import Image
from dither_pie_amd import backend
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer, prepare_palette
rs=np.random.RandomState(3)
h,w=540,960
y,x=np.mgrid[0:h,0:w]
def img(kind):
    if kind=="smooth":   # sky-like gradients + a few objects
        r=80+60*np.sin(x/300.0)+40*(y/h); g=110+50*np.cos(y/200.0)+20*np.sin(x/97.0); b=160+70*(y/h)+10*np.sin((x+y)/50.0)
    elif kind=="dark":   # mostly dark tones
        r=20+25*np.sin(x/120.0)**2+15*(y/h); g=18+22*np.cos(y/90.0)**2; b=25+30*np.sin((x+y)/150.0)**2
    else:                # saturated patches
        r=128+127*np.sign(np.sin(x/80.0))*np.abs(np.sin(y/60.0)); g=128+127*np.sin(x/40.0+y/70.0); b=128+127*np.cos(x/90.0)*np.sin(y/45.0)
    a=np.stack([r,g,b],-1)+rs.normal(0,3,(h,w,3))
    return np.clip(a,0,255).astype(np.uint8)
g=torch.Generator(device='cuda'); g.manual_seed(1234)
frames=torch.randint(0,256,(24,2160,3840,3),dtype=torch.uint8,device='cuda',generator=g); out=torch.empty_like(frames)
for kind in ("smooth","dark","patches"):
    a=img(kind)
    for K in ((16,64,256) if len(sys.argv) < 2 else [int(v) for v in sys.argv[1:]]):
        for src,pal in (("median_cut",ColorReducer.reduce_colors(Image.fromarray(a,"RGB"),K)),("kmeans",ColorReducer.generate_kmeans_palette(Image.fromarray(a,"RGB"),K,random_state=42))):
            P=backend.Palette(*prepare_palette(pal,False),accel=True)
            big=torch.from_numpy(a).cuda().repeat(4,4,1).unsqueeze(0).repeat(24,1,1,1).contiguous()   # 24 frames of 2160x3840 image-like content
            d=ImageDitherer(K,DitherMode.BAYER,pal,False,{"size":"8x8"})
            d.apply_dithering_frames(big,out=out); torch.cuda.synchronize()
            t0=time.perf_counter()
            for _ in range(3): d.apply_dithering_frames(big,out=out)
            torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/3
            d.apply_dithering_frames(frames,out=out); torch.cuda.synchronize()
            t0=time.perf_counter()
            for _ in range(3): d.apply_dithering_frames(frames,out=out)
            torch.cuda.synchronize(); dn=(time.perf_counter()-t0)/3
            print(f"{kind:8s} K={K:3d} {src:10s}: distinct {len(set(pal)):3d}, table words {P.accel_entries:6d}, max list {P.accel_max_list:3d}; image-like frames {dt*1e3:7.3f} ms, noise frames {dn*1e3:7.3f} ms", flush=True)
