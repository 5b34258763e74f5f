"""Floyd-Steinberg with few frames in flight: one workgroup per frame (DP_ED_ONE_WG=1) vs a frame's bands spread over
several workgroups (default when the batch leaves CUs idle).  Prints times and checks both give the same bytes."""
import os, sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
g = torch.Generator(device='cuda'); g.manual_seed(5)
frames = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g)
outs = {}
for variant, K in (("floyd_steinberg", 16), ("floyd_steinberg", 256), ("jjn", 16), ("atkinson", 16)):
    pal = ColorReducer.generate_uniform_palette(K) if K == 16 else [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (K, 3))]
    d = ImageDitherer(K, DitherMode.ERROR_DIFFUSION, pal, False, {"variant": variant, "serpentine": "false"})
    for nf in (1, 4, 24):
        res = []
        for one in ("1", ""):
            if one: os.environ["DP_ED_ONE_WG"] = one
            else: os.environ.pop("DP_ED_ONE_WG", None)
            out = torch.empty_like(frames[:nf])
            d.apply_dithering_frames(frames[:nf], out=out); torch.cuda.synchronize()
            t0 = time.perf_counter(); d.apply_dithering_frames(frames[:nf], out=out); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            res.append((dt, out))
        same = bool(torch.equal(res[0][1], res[1][1]))
        print(f"{variant:16s} K={K:3d} frames={nf:2d}: one workgroup per frame {res[0][0]*1e3:8.2f} ms, spread {res[1][0]*1e3:8.2f} ms, same bytes {same}", flush=True)
