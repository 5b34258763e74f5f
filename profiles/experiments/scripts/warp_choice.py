"""Which cell table serves image-derived palettes best: times the image-like frames with DP_FORCE_TABLE = u4/u8/w4/w8
(experiment behind the accelerator's choice of table; prints the accelerator's own estimates next to the timings)."""
import os; os.environ.setdefault("DITHER_PIE_EXPERIMENTS", "1")  # the DP_* switches live in libditherpie_hip_exp.so
import os, sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from PIL import Image
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
from dither_pie_amd import dithering_lib as _dl
rs=np.random.RandomState(3)
h,w=540,960
y,x=np.mgrid[0:h,0:w]
def img(kind):
    if kind=="smooth":
        r=80+60*np.sin(x/300.0)+40*(y/h); g=110+50*np.cos(y/200.0)+20*np.sin(x/97.0); b=160+70*(y/h)+10*np.sin((x+y)/50.0)
    elif kind=="dark":
        r=20+25*np.sin(x/120.0)**2+15*(y/h); g=18+22*np.cos(y/90.0)**2; b=25+30*np.sin((x+y)/150.0)**2
    else:
        r=128+127*np.sign(np.sin(x/80.0))*np.abs(np.sin(y/60.0)); g=128+127*np.sin(x/40.0+y/70.0); b=128+127*np.cos(x/90.0)*np.sin(y/45.0)
    a=np.stack([r,g,b],-1)+rs.normal(0,3,(h,w,3))
    return np.clip(a,0,255).astype(np.uint8)
out=None
for kind in ("smooth","dark","patches"):
    a=img(kind)
    big=torch.from_numpy(a).cuda().repeat(4,4,1).unsqueeze(0).repeat(24,1,1,1).contiguous()
    if out is None: out=torch.empty_like(big)
    for K in (8,16,32,64,128,256):
        for src in ("median_cut","kmeans"):
            pal=ColorReducer.reduce_colors(Image.fromarray(a,"RGB"),K) if src=="median_cut" else ColorReducer.generate_kmeans_palette(Image.fromarray(a,"RGB"),K,random_state=42)
            res=[]
            for force in ("","u4","u8","w4","w8"):
                if force in ("u4","w4") and K>64: res.append("   -  "); continue
                os.environ["DP_FORCE_TABLE"]=force
                if not force: os.environ["DP_DEBUG_ACCEL"]="1"
                else: os.environ.pop("DP_DEBUG_ACCEL",None)
                _dl._PALETTES.clear()  # the device palette (and its table) is cached per palette: rebuild it
                d=ImageDitherer(K,DitherMode.BAYER,pal,False,{"size":"8x8"})
                for _ in range(3): d.apply_dithering_frames(big,out=out)
                torch.cuda.synchronize()
                t0=time.perf_counter()
                for _ in range(3): d.apply_dithering_frames(big,out=out)
                torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/3
                res.append(f"{dt*1e3:6.3f}")
            print(f"{kind:8s} K={K:3d} {src:10s}: auto {res[0]}  u4 {res[1]}  u8 {res[2]}  w4 {res[3]}  w8 {res[4]} ms", flush=True)
