"""Kernel time of one ordered mode on the C2 batch, for A/B comparisons of two library builds in ONE gpurun call
(box-to-box variance is +-5 %): run it alternately with DP_LIB_PATH pointing at either build.
usage: ab_kernel.py <label> [bayer8|bayer4|blue|ign|none|bayer8g (use_gamma)] [launches]"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
from dither_pie_amd import backend as be
label = sys.argv[1]
which = sys.argv[2] if len(sys.argv) > 2 else "bayer8"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
pal = [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (256, 3))]
f = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (24, 2160, 3840, 3), dtype=np.uint8)).cuda(); o = torch.empty_like(f)
mode, params = {"none": (DitherMode.NONE, {}), "bayer8": (DitherMode.BAYER, {"size": "8x8"}), "bayer4": (DitherMode.BAYER, {"size": "4x4"}),
                "blue": (DitherMode.BLUE_NOISE, {"size": 64}), "ign": (DitherMode.INTERLEAVED_GRADIENT_NOISE, {}), "bayer8g": (DitherMode.BAYER, {"size": "8x8"})}[which]
gamma = which.endswith("g")
d = ImageDitherer(256, mode, pal, gamma, params).prepare()
for _ in range(5): d.apply_dithering_frames(f, out=o)
ts = []
for _ in range(n):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); d.apply_dithering_frames(f, out=o); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
import hashlib
print(f"{label:8s} {which}: min {ts[0]:.4f} ms  median {ts[len(ts)//2]:.4f} ms  out sha {hashlib.sha256(o[:2].cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
