"""C5's launch (100 frames of 1080p noise, Bayer 4x4, 16 uniform colours) on the experiments library with and without a DP_*
switch, alternating in one process; outputs compared.  usage: c5_time.py [SWITCH=DP_LEAN_NO_HALF] [K=16] [bayer4|bayer8|none|ign]"""
import os, sys; sys.path.insert(0, '.')
os.environ["DITHER_PIE_EXPERIMENTS"] = "1"
import torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
from dither_pie_amd import backend
SWITCH = sys.argv[1] if len(sys.argv) > 1 else "DP_LEAN_NO_HALF"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
which = sys.argv[3] if len(sys.argv) > 3 else "bayer4"
mode, params = {"none": (DitherMode.NONE, {}), "bayer8": (DitherMode.BAYER, {"size": "8x8"}), "bayer4": (DitherMode.BAYER, {"size": "4x4"}),
                "ign": (DitherMode.INTERLEAVED_GRADIENT_NOISE, {})}[which]
g = torch.Generator(device='cuda'); g.manual_seed(1)
f = torch.randint(0, 256, (100, 1080, 1920, 3), dtype=torch.uint8, device='cuda', generator=g)
outs = [torch.empty_like(f), torch.empty_like(f)]
d = ImageDitherer(K, mode, ColorReducer.generate_uniform_palette(K), False, params).prepare()
for rep in range(3):
    for i, env in enumerate(({}, {SWITCH: "1"})):
        for k, v in env.items(): os.environ[k] = v
        for _ in range(5): d.apply_dithering_frames(f, out=outs[i])
        ts = []
        for _ in range(20):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); d.apply_dithering_frames(f, out=outs[i]); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        backend.profile_enable(True); d.apply_dithering_frames(f, out=outs[i]); torch.cuda.synchronize(); pr = backend.profile_read(); backend.profile_enable(False)
        print(f"{'default' if not env else SWITCH + '=1':22s} kernel {pr[0]:.4f} ms, call min {min(ts):.4f} median {sorted(ts)[10]:.4f} ms", flush=True)
        for k in env: del os.environ[k]
print("identical bytes:", torch.equal(outs[0], outs[1]))
