"""Per-band timestamps of ONE 4K Floyd-Steinberg frame, from a twin whose ed_wavefront_kernel (few-frames instances) stores
{band start, first boundary fetch, band end} (wall_clock64, 100 MHz) over the first 24 output bytes of every band's first row
(wrong pixels there; patch in profiles/experiments/r05_priced_structures.md section 7).  usage: ed_band_trace.py [variant] [K]"""
import os, sys
sys.path.insert(0, '.')
os.environ["DITHER_PIE_EXPERIMENTS"] = "1"
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
variant = sys.argv[1] if len(sys.argv) > 1 else "floyd_steinberg"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
pal = ColorReducer.generate_uniform_palette(K) if K <= 64 else [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (K, 3))]
d = ImageDitherer(K, DitherMode.ERROR_DIFFUSION, pal, False, {"variant": variant, "serpentine": "false"})
g = torch.Generator(device='cuda'); g.manual_seed(1)
f = torch.randint(0, 256, (1, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
for _ in range(3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); d.apply_dithering_frames(f, out=o); e1.record(); torch.cuda.synchronize()
print(f"{variant} K={K}: {e0.elapsed_time(e1):.3f} ms")
rows = o[0].cpu().numpy()
st = np.array([np.frombuffer(rows[b * 64, :8].tobytes(), np.uint64) for b in range(34)]).astype(np.int64)
t0 = st[:, 0].min()
us = (st - t0) / 100.0
print("band  start     go       end     go-prev_go  end-prev_end  (us)")
for b in range(34):
    print(f"{b:3d} {us[b,0]:8.1f} {us[b,1] if st[b,1] else 0:8.1f} {us[b,2]:8.1f}   {(us[b,1]-us[b-1,1]) if b > 1 else 0:8.1f} {(us[b,2]-us[b-1,2]) if b else 0:8.1f}")
