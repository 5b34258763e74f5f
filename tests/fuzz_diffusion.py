"""Random error-diffusion cases (all eight tap sets, both scans, both arithmetics, the four variable-coefficient diffusers and
the hybrid diffuser's numba branch)
against the oracle: palette sizes around every table boundary (2, 8, 9, 16, 17, 64, 256), random / uniform / clustered
palettes, gamma on and off, shapes from 1x1 up to a few bands, batches.  run(seed, n) -> number of mismatches.
Used by tests/test_gpu_kernels.py; `python tests/fuzz_diffusion.py [seed] [n] [K,K,...]` runs it by hand on a GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

VARIANTS = ["floyd_steinberg", "jjn", "stucki", "burkes", "atkinson", "sierra", "sierra_two_row", "sierra_lite"]
VAR_MODES = ["perceptual", "hybrid", "adaptive_variance", "ostromoukhov"]


def _palette(rs, orc, K, kind):
    if kind == 0:
        return orc.generate_uniform_palette(K)
    if kind == 1:
        return [tuple(int(v) for v in c) for c in rs.randint(0, 256, (K, 3))]
    base = rs.randint(20, 230, 3)  # clustered: long candidate lists, refined cells
    return [tuple(int(v) for v in np.clip(base + rs.randint(-12, 13, 3), 0, 255)) for _ in range(K)]


def _image(rs, h, w, kind, pal):
    if kind == 0:
        return rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    if kind == 1:  # smooth ramp + grain
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([xx * 3 + yy, yy * 2 + 40, (xx + yy) * 2], -1) % 256
        return np.clip(a + rs.randint(-3, 4, (h, w, 3)), 0, 255).astype(np.uint8)
    pa = np.asarray(pal, np.int64)  # tie-rich: midpoints of palette entries, flat patches
    i, j = rs.randint(0, len(pal), (2, h, w))
    return ((pa[i] + pa[j]) // 2).astype(np.uint8)


def run(seed, n, verbose=False, ks=None):
    import torch
    from dither_pie_amd import backend as be
    from oracle import oracle as orc
    rs = np.random.RandomState(seed)
    bad = 0
    for case in range(n):
        K = int(rs.choice(ks if ks else [2, 5, 8, 9, 12, 16, 17, 40, 64, 256]))
        pal = _palette(rs, orc, K, int(rs.randint(0, 3)))
        h = int(rs.choice([1, 2, 3, 17, 63, 64, 65, 130, 200]))
        w = int(rs.choice([1, 2, 3, 7, 8, 9, 31, 64, 97, 160, 333]))
        gamma = bool(rs.randint(0, 4) == 0)
        nf = int(rs.choice([1, 1, 2, 5]))
        frames = np.stack([_image(rs, h, w, int(rs.randint(0, 3)), pal) for _ in range(nf)])
        pal_f32, out_colors, lut_in = orc.prepare_palette(pal, gamma)
        P = be.Palette(pal_f32, out_colors, lut_in)
        t = torch.from_numpy(frames).cuda()
        which = int(rs.randint(0, 4))
        # several frames per workgroup (the persistent grid of batches larger than the device; libditherpie_hip_exp.so reads the switch)
        grid = int(rs.randint(1, nf + 1)) if nf > 1 and rs.randint(0, 2) else 0
        if grid:
            os.environ["DP_ED_GRID"] = str(grid)
        else:
            os.environ.pop("DP_ED_GRID", None)
        if which < 3:
            variant = VARIANTS[int(rs.randint(0, len(VARIANTS)))]
            serp = bool(rs.randint(0, 2))
            arith = "numba" if which == 2 else "python"
            taps, div = orc.ed_kernel(variant)
            out = be.error_diffusion(t, P, taps, div, serp, arithmetic=arith).cpu().numpy()
            fn = orc.error_diffusion_numba_u8 if arith == "numba" else orc.error_diffusion_u8
            ref = np.stack([fn(f, pal_f32, out_colors, lut_in, variant, serp) for f in frames])
            what = f"{arith} {variant} serp={serp}"
        else:
            mode = VAR_MODES[int(rs.randint(0, 4))]
            params = {"serpentine": "true"} if (mode == "ostromoukhov" and rs.randint(0, 2)) else {}
            from dither_pie_amd.dithering_lib import DitherMode, ImageDitherer
            if mode == "hybrid" and rs.randint(0, 2):   # the hybrid diffuser's own numba branch (_hybrid_numba)
                lf, cf = float(rs.choice([1.0, 1.4, 0.3])), float(rs.choice([0.2, 0.0, 1.0]))
                out = be.hybrid_numba(t, P, lf, cf).cpu().numpy()
                ref = np.stack([orc.hybrid_numba_u8(f, pal_f32, out_colors, lut_in, lf, cf) for f in frames])
                what = f"hybrid numba {lf} {cf}"
            else:
                d = ImageDitherer(K, DitherMode(mode), pal, gamma, params)
                out = d.apply_dithering_frames(t).cpu().numpy()
                ref = np.stack([orc.apply_dithering(f, pal, mode, params, gamma) for f in frames])
                what = f"{mode} {params}"
        os.environ.pop("DP_ED_GRID", None)
        if not np.array_equal(out, ref):
            bad += 1
            print(f"MISMATCH seed={seed} case={case}: {what} K={K} {nf}x{h}x{w} gamma={gamma} grid={grid}: "
                  f"{int((out != ref).any(-1).sum())} pixels", flush=True)
        elif verbose:
            print(f"ok {case}: {what} K={K} {nf}x{h}x{w} gamma={gamma}", flush=True)
    return bad


if __name__ == "__main__":
    s = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    # a third argument "K,K,...": palette sizes to draw from (e.g. 9,12,16: the expanded-key scan of ed_nearest.hip.h); anything else: verbose
    ks = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 and sys.argv[3][0].isdigit() else None
    print("mismatches:", run(s, n, verbose=len(sys.argv) > 3 and ks is None, ks=ks))
