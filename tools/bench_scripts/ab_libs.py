"""A/B of two BUILDS in one process: libditherpie_hip.so (A) against libditherpie_hip_exp.so (B), alternating, for a kernel change
that was compiled into only one of them (build A from the committed sources, apply the change, `make ../libditherpie_hip_exp.so`).
Box-to-box spread on this pool is +-4 %: differences smaller than that can only be seen this way.
usage: ab_libs.py [case ...]   cases: variant:K:frames  (e.g. floyd_steinberg:16:1 atkinson:16:1 floyd_steinberg:256:1 floyd_steinberg:16:256)
       or ordered modes on 1080p frames: bayer4:K:frames, bayer8:K:frames, none:K:frames, ign:K:frames (bayer4:16:100 = C5's launch;
       K <= 64: the uniform palette, else palr(K, 7); the cell table is built before the timing)"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import _lib, dithering_lib
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
cases = sys.argv[1:] or ["floyd_steinberg:16:1", "atkinson:16:1", "floyd_steinberg:256:1", "floyd_steinberg:16:256"]
g = torch.Generator(device='cuda'); g.manual_seed(1)
ORDERED = {"none": (DitherMode.NONE, {}), "bayer8": (DitherMode.BAYER, {"size": "8x8"}), "bayer4": (DitherMode.BAYER, {"size": "4x4"}),
           "ign": (DitherMode.INTERLEAVED_GRADIENT_NOISE, {})}
nmax = max(int(c.split(":")[2]) for c in cases)
small = all(c.split(":")[0] in ORDERED for c in cases)
f = torch.randint(0, 256, (nmax, 1080, 1920, 3) if small else (nmax, 2160, 3840, 3), dtype=torch.uint8, device='cuda', generator=g)
outs = {False: torch.empty_like(f), True: torch.empty_like(f)}
for case in cases:
    variant, K, nf = case.split(":"); K = int(K); nf = int(nf)
    pal = ColorReducer.generate_uniform_palette(K) if K <= 64 else [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (K, 3))]
    best = {False: [], True: []}
    for rep in range(3):
        for exp in (False, True):
            _lib.select(exp)
            dithering_lib.drop_device_caches()
            if variant in ORDERED:
                d = ImageDitherer(K, ORDERED[variant][0], pal, False, ORDERED[variant][1]).prepare()
            elif variant in ("perceptual", "hybrid", "adaptive_variance", "ostromoukhov"):
                d = ImageDitherer(K, DitherMode(variant), pal, False, {})
            else:
                d = ImageDitherer(K, DitherMode.ERROR_DIFFUSION, pal, False, {"variant": variant, "serpentine": "false"})
            for _ in range(2): d.apply_dithering_frames(f[:nf], out=outs[exp][:nf])
            torch.cuda.synchronize()
            ts = []
            for _ in range(12 if small else 5):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); d.apply_dithering_frames(f[:nf], out=outs[exp][:nf]); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            best[exp].append(min(ts))
            del d
    a, b = min(best[False]), min(best[True])
    same = torch.equal(outs[False][:nf], outs[True][:nf])
    print(f"{case:28s} A (product) {a:8.3f} ms   B (experiments twin) {b:8.3f} ms   B/A {b / a:.3f}   identical bytes: {same}"
          f"   [A runs {' '.join('%.3f' % v for v in best[False])} | B runs {' '.join('%.3f' % v for v in best[True])}]", flush=True)
