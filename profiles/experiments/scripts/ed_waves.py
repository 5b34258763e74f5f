"""Waves per workgroup (DP_ED_WAVES, experiments library) on a 768-frame 4K batch of Floyd-Steinberg -- the persistent grid: 16 waves 184 Gpx/s,
14: 174, 12: 169, 10: 148, 8: 130 (one MI355X).  usage: ed_waves.py"""
import os, sys; sys.path.insert(0, '.')
os.environ["DITHER_PIE_EXPERIMENTS"] = "1"
import torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
g = torch.Generator(device='cuda'); g.manual_seed(1)
nf = 768
f = torch.randint(0, 256, (nf, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
d = ImageDitherer(16, DitherMode.ERROR_DIFFUSION, ColorReducer.generate_uniform_palette(16), False, {"variant": "floyd_steinberg", "serpentine": "false"})
for rep in range(2):
    for wv in ("16", "14", "12", "10", "8"):
        os.environ["DP_ED_WAVES"] = wv
        d.apply_dithering_frames(f, out=o); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); d.apply_dithering_frames(f, out=o); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        print(f"waves {wv}: {min(ts):.2f} ms  {nf*2160*3840/min(ts)/1e6:.1f} Gpx/s", flush=True)
