"""C2-shaped timing of the ordered kernels (24 x 4K noise frames, 256 random colours) through the library's own HIP events:
the fast kernel (and its measurement variants DP_FAST_DBG=1..3) against the lean kernel, interleaved in one process."""
import os; os.environ.setdefault("DITHER_PIE_EXPERIMENTS", "1")  # the DP_* switches live in libditherpie_hip_exp.so
import os, sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import backend as be
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
pal = [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (256, 3))]
g = torch.Generator(device='cuda'); g.manual_seed(1)
F = 24
f = torch.randint(0, 256, (F, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
names = sys.argv[1:] or ["none", "bayer8", "bayer4", "blue", "ign"]
allc = {"none": (DitherMode.NONE, {}), "bayer8": (DitherMode.BAYER, {"size": "8x8"}), "bayer4": (DitherMode.BAYER, {"size": "4x4"}),
        "blue": (DitherMode.BLUE_NOISE, {"size": 64}), "ign": (DitherMode.INTERLEAVED_GRADIENT_NOISE, {})}
def run(d):
    be.profile_enable(True)
    d.apply_dithering_frames(f, out=o)
    torch.cuda.synchronize()
    main_ms, fix_ms, n = be.profile_read()
    be.profile_enable(False)
    return main_ms / max(n, 1)
for name in names:
    mode, params = allc[name]
    os.environ.pop("DP_NO_FAST", None)
    dfast = ImageDitherer(256, mode, pal, False, params); dfast.apply_dithering_frames(f[:1], out=o[:1])
    os.environ["DP_NO_FAST"] = "1"
    dlean = ImageDitherer(256, mode, pal, False, params); dlean.apply_dithering_frames(f[:1], out=o[:1])
    os.environ.pop("DP_NO_FAST", None)
    variants = [("lean", dlean, "0")] + [("fast dbg%d" % k, dfast, str(k)) for k in (0, 1, 2, 3)]
    res = {v[0]: [] for v in variants}
    for rep in range(12):
        for vn, d, dbg in variants:
            os.environ["DP_FAST_DBG"] = dbg
            if rep == 0: run(d)
            res[vn].append(run(d))
    os.environ["DP_FAST_DBG"] = "0"
    print(f"{name:8s} " + "  ".join(f"{vn}: {sorted(v)[len(v)//2]:.4f}" for vn, v in res.items()), flush=True)
