"""4-entry tables: one 160 KB workgroup per CU (DP_LEAN_NO_HALF=1) against two 80 KB workgroups.  Same process, same box."""
import os; os.environ.setdefault("DITHER_PIE_EXPERIMENTS", "1")  # the DP_* switches live in libditherpie_hip_exp.so
import sys, os; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
from oracle import oracle as orc
def frames(n, h, w, seed):
    g = torch.Generator(device='cuda'); g.manual_seed(seed)
    return torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device='cuda', generator=g)
cases = [("C5 1080p x100 bayer4x4 uniform16", DitherMode.BAYER, {"size": "4x4"}, ColorReducer.generate_uniform_palette(16), frames(100, 1080, 1920, 1)),
         ("4K x24 bayer8x8 palr16", DitherMode.BAYER, {"size": "8x8"}, orc.palr(16, 3), frames(24, 2160, 3840, 2)),
         ("4K x24 none palr16", DitherMode.NONE, {}, orc.palr(16, 3), frames(24, 2160, 3840, 2)),
         ("4K x24 IGN palr8", DitherMode.INTERLEAVED_GRADIENT_NOISE, {}, orc.palr(8, 5), frames(24, 2160, 3840, 2)),
         ("4K x24 blue_noise palr16", DitherMode.BLUE_NOISE, {"size": 64, "seed": 42}, orc.palr(16, 3), frames(24, 2160, 3840, 2))]
for name, mode, params, pal, f in cases:
    d = ImageDitherer(len(pal), mode, pal, False, params).prepare()
    o = torch.empty_like(f)
    res = {}
    outs = {}
    for rep in range(2):
        for which in ("one", "two"):
            os.environ.pop("DP_LEAN_NO_HALF", None)
            if which == "one": os.environ["DP_LEAN_NO_HALF"] = "1"
            for _ in range(30): d.apply_dithering_frames(f, out=o)
            ts = []
            for _ in range(20):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); d.apply_dithering_frames(f, out=o); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            res.setdefault(which, []).append(sorted(ts)[len(ts) // 2])
            outs[which] = o.clone()
    same = torch.equal(outs["one"], outs["two"])
    px = f.numel() / 3
    print(f"{name:34s} one wg/CU {min(res['one']):.4f} ms  two wg/CU {min(res['two']):.4f} ms  ({px / min(res['two']) / 1e6:.0f} Gpx/s)  same bytes: {same}", flush=True)
os.environ.pop("DP_LEAN_NO_HALF", None)
