import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time, numpy as np, torch
from PIL import Image
from dither_pie_amd import dithering_lib as dl
a=np.random.RandomState(1).randint(0,256,(2160,3840,3),dtype=np.uint8)
im=Image.fromarray(a); h,w=2160,3840
pin_in,pin_out=dl._pinned_pair(h*w*4)
def T(f,n=10):
    f(); ts=[]
    for _ in range(n):
        torch.cuda.synchronize(); t=time.perf_counter(); r=f(); torch.cuda.synchronize(); ts.append((time.perf_counter()-t)*1e3)
    ts.sort(); return round(ts[n//2],2)
print("rgbx_into", T(lambda: dl._pil_rgbx_into(im, pin_in.numpy())))
print("H2D 33MB", T(lambda: pin_in.view(h,w,4).cuda(non_blocking=True)))
dev4=pin_in.view(h,w,4).cuda()
print("slice contiguous", T(lambda: dev4[...,:3].contiguous()))
d3=dev4[...,:3].contiguous()
def pack():
    o=torch.empty((h,w,4),dtype=torch.uint8,device="cuda"); o[...,:3]=d3; o[...,3]=255; return o
print("pack rgbx on gpu", T(pack))
o4=pack()
print("D2H 33MB", T(lambda: pin_out.view(h,w,4).copy_(o4,non_blocking=True)))
print("frombuffer+convert", T(lambda: Image.frombuffer("RGBX",(w,h),pin_out.numpy(),"raw","RGBX",0,1).convert("RGB")))
print("frombuffer+copy", T(lambda: Image.frombuffer("RGBX",(w,h),pin_out.numpy(),"raw","RGBX",0,1).copy()))
v=np.empty((h,w,4),np.uint8)
print("frombuffer+convert (pageable src)", T(lambda: Image.frombuffer("RGBX",(w,h),v.reshape(-1),"raw","RGBX",0,1).convert("RGB")))
