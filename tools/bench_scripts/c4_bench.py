import sys, time; sys.path.insert(0,'.')
import torch, numpy as np
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer, generate_blue_noise
from dither_pie_amd import backend, kmeans
def T(fn, n=3):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
g=torch.Generator(device='cuda'); g.manual_seed(99)
img=torch.randint(0,256,(4320,7680,3),dtype=torch.uint8,device='cuda',generator=g)
cent=torch.rand(32,3,dtype=torch.float64,device='cuda')*255
dt=T(lambda: backend.kmeans_step(img,cent)); print(f"kmeans_step 8K K=32: {dt*1e3:.3f} ms  {img.numel()/dt/1e9:.1f} GB/s read")
t0=time.perf_counter(); pal,c,inertia,it=kmeans.fit_palette(img.reshape(-1,3),32); torch.cuda.synchronize(); print(f"fit_palette 8K K=32: {time.perf_counter()-t0:.3f} s, {it} iterations, inertia {inertia:.4g}")
for size in (64,128):
    backend.Thresholds.blue_noise(size, 1)  # warm
    t0=time.perf_counter(); backend.Thresholds.blue_noise(size, 42); print(f"blue_noise({size},42) on device: {time.perf_counter()-t0:.3f} s")
d=ImageDitherer(32, DitherMode.BLUE_NOISE, pal, False, {"size":64,"seed":42})
o=torch.empty_like(img)
dt=T(lambda: d.apply_dithering_frames(img,out=o)); print(f"blue-noise dither 8K K=32: {dt*1e3:.3f} ms  {img.numel()/3/dt/1e9:.1f} Gpx/s")
t0=time.perf_counter(); P=backend.Palette(*__import__('dither_pie_amd.dithering_lib',fromlist=['x']).prepare_palette([tuple(int(v) for v in c) for c in np.random.RandomState(5).randint(0,256,(256,3))],False)); torch.cuda.synchronize(); print(f"palette create (K=256, accelerator build): {time.perf_counter()-t0:.4f} s; table words {P.accel_entries}")
