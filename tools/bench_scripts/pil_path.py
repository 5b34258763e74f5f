import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from PIL import Image
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
pal=[tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0,256,(256,3))]
img=Image.fromarray(np.random.RandomState(1).randint(0,256,(2160,3840,3),dtype=np.uint8),"RGB")
d=ImageDitherer(256, DitherMode.BAYER, pal, False, {"size":"8x8"})
d.apply_dithering(img); d.apply_dithering(img)
def T(f,n=5):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): r=f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
print("apply_dithering(PIL 4K) total: %.2f ms"%T(lambda: d.apply_dithering(img)))
arr=np.array(img.convert("RGB"),dtype=np.uint8)
print("  image.convert+np.array: %.2f ms"%T(lambda: np.array(img.convert("RGB"),dtype=np.uint8)))
print("  from_numpy().cuda() (pageable H2D 24.9MB): %.2f ms"%T(lambda: torch.from_numpy(arr).cuda()))
x=torch.from_numpy(arr).cuda()
print("  apply_dithering_frames: %.2f ms"%T(lambda: d.apply_dithering_frames(x)))
o=d.apply_dithering_frames(x)
print("  out.cpu().numpy() (D2H): %.2f ms"%T(lambda: o.cpu().numpy()))
on=o.cpu().numpy()
print("  Image.fromarray: %.2f ms"%T(lambda: Image.fromarray(on,"RGB")))
pin=torch.empty_like(x,device='cpu').pin_memory()
def viapin():
    pin.numpy()[...]=arr; return pin.cuda(non_blocking=True)
print("  H2D via pinned staging (memcpy+async): %.2f ms"%T(viapin))
pout=torch.empty_like(x,device='cpu').pin_memory()
def d2hpin():
    pout.copy_(o,non_blocking=True); torch.cuda.synchronize(); return pout.numpy()
print("  D2H into pinned: %.2f ms"%T(d2hpin))
