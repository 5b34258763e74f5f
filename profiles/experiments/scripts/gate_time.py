"""Adaptive-variance diffuser at 4K, 256 frames: the variance gate and the diffusion timed separately (events)."""
import os; os.environ.setdefault("DITHER_PIE_EXPERIMENTS", "1")  # the DP_* switches live in libditherpie_hip_exp.so
import os, sys; sys.path.insert(0, '.')
import torch
from dither_pie_amd import backend as be
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
pal = ColorReducer.generate_uniform_palette(16)
g = torch.Generator(device='cuda'); g.manual_seed(1)
f = torch.randint(0, 256, (256, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
d = ImageDitherer(16, DitherMode.ADAPTIVE_VARIANCE, pal, False, {})
d.apply_dithering_frames(f, out=o); torch.cuda.synchronize()
def T(fn, n=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
import numpy as np
pobj = be.Palette(np.asarray(pal, np.float32), np.asarray(pal, np.uint8), None)
print("whole call      %.2f ms" % T(lambda: d.apply_dithering_frames(f, out=o)))
if pobj is not None:
    print("variance gate   %.2f ms" % T(lambda: be.variance_gate(f, pobj, 300.0, 1)))
    os.environ["DP_GATE_TWO_PASS"] = "1"
    print("  (two passes)  %.2f ms" % T(lambda: be.variance_gate(f, pobj, 300.0, 1)))
