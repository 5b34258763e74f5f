import sys, time; sys.path.insert(0,'.')
import __graft_entry__ as g
g.smoke()
import torch, numpy as np
from dither_pie_amd import backend
from dither_pie_amd.dithering_lib import prepare_palette, ImageDitherer, DitherMode
from PIL import Image
pal=[tuple(int(v) for v in c) for c in np.random.RandomState(5).randint(0,256,(256,3))]
P=backend.Palette(*prepare_palette(pal,False)); torch.cuda.synchronize()
t0=time.perf_counter(); P.build_accel(); torch.cuda.synchronize(); print(f"accelerator build K=256: {(time.perf_counter()-t0)*1e3:.1f} ms, table words {P.accel_entries}, longest list {P.accel_max_list}")
pal16=[tuple(int(v) for v in c) for c in np.random.RandomState(6).randint(0,256,(16,3))]
P=backend.Palette(*prepare_palette(pal16,False)); t0=time.perf_counter(); P.build_accel(); torch.cuda.synchronize(); print(f"accelerator build K=16: {(time.perf_counter()-t0)*1e3:.1f} ms")
img=Image.fromarray(np.random.RandomState(1).randint(0,256,(2160,3840,3),dtype=np.uint8))
d=ImageDitherer(256, DitherMode.BAYER, pal, False, {"size":"8x8"})
d.apply_dithering(img)
t0=time.perf_counter(); d.apply_dithering(img); print(f"apply_dithering(PIL 4K, K=256) end to end incl. PCIe both ways: {(time.perf_counter()-t0)*1e3:.1f} ms")
small=Image.fromarray(np.random.RandomState(1).randint(0,256,(512,512,3),dtype=np.uint8))
d16=ImageDitherer(16, DitherMode.BAYER, pal16, False, {"size":"4x4"}); d16.apply_dithering(small)
t0=time.perf_counter(); d16.apply_dithering(small); print(f"apply_dithering(PIL 512x512, K=16) (config 1 shape): {(time.perf_counter()-t0)*1e3:.2f} ms")
