"""One 4K frame (and 24, and 256) through error diffusion, timed with events over several repetitions.
usage: ed_single.py [variant | perceptual | hybrid | adaptive_variance | ostromoukhov] [K] [reps] [nmax=256] [H=2160] [W=3840]
(nmax: the largest batch; above the number of CUs the diffusion kernel runs its persistent grid)"""
import sys; sys.path.insert(0, '.')
import torch, numpy as np
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
variant = sys.argv[1] if len(sys.argv) > 1 else "floyd_steinberg"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
pal = ColorReducer.generate_uniform_palette(K) if K <= 64 else [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (K, 3))]
g = torch.Generator(device='cuda'); g.manual_seed(1)
nmax = int(sys.argv[4]) if len(sys.argv) > 4 else 256
H = int(sys.argv[5]) if len(sys.argv) > 5 else 2160
W = int(sys.argv[6]) if len(sys.argv) > 6 else 3840
f = torch.randint(0, 256, (nmax, H, W, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
if variant in ("perceptual", "hybrid", "adaptive_variance", "ostromoukhov"):
    d = ImageDitherer(K, DitherMode(variant), pal, False, {})
else:
    d = ImageDitherer(K, DitherMode.ERROR_DIFFUSION, pal, False, {"variant": variant, "serpentine": "false"})
for nf in sorted({1, 24, min(256, nmax), nmax}):
    d.apply_dithering_frames(f[:nf], out=o[:nf]); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); d.apply_dithering_frames(f[:nf], out=o[:nf]); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{variant} K={K} frames={nf}: min {min(ts):.2f} ms  median {sorted(ts)[len(ts)//2]:.2f} ms  {nf*H*W/min(ts)/1e6:.1f} Gpx/s", flush=True)
