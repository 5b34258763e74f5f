"""use_gamma (float palettes), 24 4K frames: ordered_compact_float_kernel (default) against ordered_lean_float_kernel
(DP_NO_COMPACT_KERNEL=1); whole call (main kernel + fix-up pass), same
process, outputs compared byte for byte.  usage: gamma_ab.py [SWITCH]   (another DP_* switch to set for the second
leg instead: how profiles/experiments/r04_rotated_records.md was measured, with a switch that variant carried)"""
import os, sys; sys.path.insert(0, '.')
os.environ["DITHER_PIE_EXPERIMENTS"] = "1"
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
from dither_pie_amd import backend
SWITCH = sys.argv[1] if len(sys.argv) > 1 else "DP_NO_COMPACT_KERNEL"
def palr(K, seed=7): return [tuple(int(v) for v in c) for c in np.random.RandomState(seed).randint(0, 256, (K, 3))]
g = torch.Generator(device='cuda'); g.manual_seed(1234)
f = torch.randint(0, 256, (24, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g)
def timeit(fn, n=8):
    for _ in range(5): fn()
    ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
outs = [torch.empty_like(f) for _ in range(2)]
for K in (256, 64, 16):
    for mode, params in ((DitherMode.BAYER, {"size": "8x8"}), (DitherMode.NONE, {}), (DitherMode.INTERLEAVED_GRADIENT_NOISE, {})):
        d = ImageDitherer(K, mode, palr(K), True, params).prepare()
        res = []; kms = []
        for i, env in enumerate(({}, {SWITCH: "1"})):
            for k, v in env.items(): os.environ[k] = v
            res.append(timeit(lambda: d.apply_dithering_frames(f, out=outs[i])))
            backend.profile_enable(True); d.apply_dithering_frames(f, out=outs[i]); torch.cuda.synchronize()
            m, fx, n = backend.profile_read(); backend.profile_enable(False); kms.append((m, fx))
            for k in env: del os.environ[k]
        same = torch.equal(outs[0], outs[1])
        print(f"gamma K={K:3d} {mode.value:6s}: compact {res[0]:.3f} ms (kernel {kms[0][0]:.3f} + fix-up {kms[0][1]:.3f}; {24*2160*3840/res[0]/1e6:.1f} Gpx/s) | "
              f"lean float {res[1]:.3f} ms (kernel {kms[1][0]:.3f}) | identical: {same}", flush=True)
