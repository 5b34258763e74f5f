"""The variable-weight diffusers on one 4K frame / 24 frames: one workgroup per frame (DP_ED_ONE_WG=1) vs spread."""
import os, sys, time; sys.path.insert(0, '.')
import torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
pal = ColorReducer.generate_uniform_palette(16)
g = torch.Generator(device='cuda'); g.manual_seed(1)
f = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g)
for mode in (DitherMode.PERCEPTUAL, DitherMode.HYBRID, DitherMode.ADAPTIVE_VARIANCE, DitherMode.OSTROMOUKHOV):
    d = ImageDitherer(16, mode, pal, False, {})
    for nf in (1, 24):
        res = []
        for one in ("1", ""):
            if one: os.environ["DP_ED_ONE_WG"] = one
            else: os.environ.pop("DP_ED_ONE_WG", None)
            o = torch.empty_like(f[:nf])
            d.apply_dithering_frames(f[:nf], out=o); torch.cuda.synchronize()
            t0 = time.perf_counter(); d.apply_dithering_frames(f[:nf], out=o); torch.cuda.synchronize(); res.append((time.perf_counter() - t0, o))
        print(f"{mode.value:18s} frames={nf:2d}: one workgroup per frame {res[0][0]*1e3:7.2f} ms, spread {res[1][0]*1e3:7.2f} ms, same bytes {bool(torch.equal(res[0][1], res[1][1]))}", flush=True)
