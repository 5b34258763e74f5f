import sys; sys.path.insert(0,'.')
import numpy as np
from PIL import Image
from dither_pie_amd import backend
from dither_pie_amd.dithering_lib import prepare_palette, ColorReducer
rs=np.random.RandomState(3)
h,w=540,960
y,x=np.mgrid[0:h,0:w]
r=80+60*np.sin(x/300.0)+40*(y/h); g=110+50*np.cos(y/200.0)+20*np.sin(x/97.0); b=160+70*(y/h)+10*np.sin((x+y)/50.0)
a=np.clip(np.stack([r,g,b],-1)+rs.normal(0,3,(h,w,3)),0,255).astype(np.uint8)
r=20+25*np.sin(x/120.0)**2+15*(y/h); g=18+22*np.cos(y/90.0)**2; b=25+30*np.sin((x+y)/150.0)**2
d=np.clip(np.stack([r,g,b],-1)+rs.normal(0,3,(h,w,3)),0,255).astype(np.uint8)
for name,im in (("smooth",a),("dark",d)):
    for K in (64,128,256):
        pal=ColorReducer.reduce_colors(Image.fromarray(im,"RGB"),K)
        print(name,K,file=sys.stderr)
        try: P=backend.Palette(*prepare_palette(pal,False),accel=True)
        except Exception as e: print("ERR",e,file=sys.stderr)
