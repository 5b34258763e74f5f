#!/usr/bin/env python3
"""Generate the golden fixtures from the REFERENCE itself (build container only).

Run:  python tests/golden/make_golden.py            (needs /root/reference; ~3-4 min)
      python tests/golden/make_golden.py --append   (only the cases kat.json does not hold yet)
      python tests/golden/make_golden.py --numba    (on a host where `import numba` works: records the reference's numba
                                                     branches into numba.json / numba.npz -- pins SURVEY row a7; a no-op with
                                                     a message where numba is missing, as in the build container)

Imports dobrosketchkun/dither_pie's dithering_lib from /root/reference (read-only; an
in-memory stub stands in for the unused `pywt` import, nothing is written there) and
records, for seeded synthetic inputs (formulas in oracle/oracle.py: rnd/grad/palr),
the outputs of ImageDitherer.apply_dithering and friends.  Only DATA is stored:
  kat.json     sha256[:16] of inputs/outputs for the known-answer tests (SURVEY.md App. B + more)
  small.npz    full small arrays: threshold tables, blue-noise matrices, gamma LUTs,
               IGN thresholds, small dither outputs, cKDTree structures and tie queries,
               k-means fixtures
Versions are recorded in kat.json["versions"].
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DITHER_PIE_REFERENCE", "/root/reference")

sys.modules.setdefault("pywt", types.ModuleType("pywt"))
sys.path.insert(0, REF)
import dithering_lib as dl  # noqa: E402  (the reference)
from PIL import Image  # noqa: E402

sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.oracle import grad, imgl, palr, rnd  # noqa: E402  (input formulas only)


def H(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def make_input(spec):
    kind = spec[0]
    if kind == "rnd":
        return rnd(spec[1], spec[2], spec[3])
    if kind == "grad":
        return grad(spec[1], spec[2])
    if kind == "imgl":
        return imgl(spec[1], spec[2], spec[3], spec[4])
    raise ValueError(spec)


def make_palette(spec):
    kind = spec[0]
    if kind == "U":
        return dl.ColorReducer.generate_uniform_palette(spec[1])
    if kind == "palr":
        return palr(spec[1], spec[2] if len(spec) > 2 else 7)
    if kind == "list":
        return [tuple(c) for c in spec[1]]
    raise ValueError(spec)


def run_ref(arr, pal, mode, params, gamma):
    d = dl.ImageDitherer(len(pal), dl.DitherMode(mode), list(pal), gamma, dict(params))
    return np.array(d.apply_dithering(Image.fromarray(arr)))


ED_VARIANTS = ["floyd_steinberg", "jjn", "stucki", "burkes", "atkinson", "sierra", "sierra_two_row",
               "sierra_lite"]

# (name, mode, params, palette spec, input spec, gamma, keep_full_output)
CASES = [
    ("bayer4_U16_rnd512", "bayer", {"size": "4x4"}, ("U", 16), ("rnd", 512, 512, 1234), False, False),
    ("bayer4_U16_grad", "bayer", {"size": "4x4"}, ("U", 16), ("grad", 333, 500), False, True),
    ("bayer2_p32_grad", "bayer", {"size": "2x2"}, ("palr", 32), ("grad", 333, 500), False, False),
    ("bayer8_p32_grad", "bayer", {"size": "8x8"}, ("palr", 32), ("grad", 333, 500), False, True),
    ("bayer16_p32_grad", "bayer", {"size": "16x16"}, ("palr", 32), ("grad", 333, 500), False, False),
    ("bayerpsx_p32_grad", "bayer", {"size": "psx4x4"}, ("palr", 32), ("grad", 333, 500), False, False),
    ("bayerdefault_p32_grad", "bayer", {}, ("palr", 32), ("grad", 97, 131), False, True),
    ("bayerbogus_p32_grad", "bayer", {"size": "nope"}, ("palr", 32), ("grad", 97, 131), False, True),
    ("none_p256_grad1080", "none", {}, ("palr", 256), ("grad", 1080, 1920), False, False),
    ("bayer8_p256_grad1080", "bayer", {"size": "8x8"}, ("palr", 256), ("grad", 1080, 1920), False, False),
    ("bayer8_p256_rnd4k", "bayer", {"size": "8x8"}, ("palr", 256), ("rnd", 2160, 3840, 1234), False, False),
    ("none_p256_rnd_small", "none", {}, ("palr", 256), ("rnd", 203, 317, 11), False, True),
    ("bayer8_p256_rnd_small", "bayer", {"size": "8x8"}, ("palr", 256), ("rnd", 203, 317, 11), False, True),
    ("bayer8_p256_grad_small", "bayer", {"size": "8x8"}, ("palr", 256), ("grad", 203, 317), False, True),
    ("bayer4_p100_grad_small", "bayer", {"size": "4x4"}, ("palr", 100, 3), ("grad", 203, 317), False, True),
    ("none_U16_grad_small", "none", {}, ("U", 16), ("grad", 203, 317), False, True),
    ("none_p7_grad_small", "none", {}, ("palr", 7, 5), ("grad", 203, 317), False, True),
    ("bayer4_p7_grad_small", "bayer", {"size": "4x4"}, ("palr", 7, 5), ("grad", 203, 317), False, True),
    ("bayer4_p2_grad_small", "bayer", {"size": "4x4"}, ("palr", 2, 5), ("grad", 64, 96), False, True),
    ("bayer4_dup_grad_small", "bayer", {"size": "4x4"},
     ("list", [(10, 20, 30), (200, 100, 50), (10, 20, 30), (200, 100, 50), (0, 0, 0)]), ("grad", 64, 96), False, True),
    ("bayer4_U64_grad_small", "bayer", {"size": "4x4"}, ("U", 64), ("grad", 203, 317), False, True),
    ("bayer8_U256_rnd_small", "bayer", {"size": "8x8"}, ("U", 256), ("rnd", 128, 160, 21), False, True),
    ("ign_p32_grad1080", "IGN", {}, ("palr", 32), ("grad", 1080, 1920), False, False),
    ("ign_s25_p32_rnd", "IGN", {"scale": 2.5, "seed": 17}, ("palr", 32), ("rnd", 200, 300, 3), False, True),
    ("ign_p256_grad_small", "IGN", {"scale": 0.7, "seed": 3}, ("palr", 256), ("grad", 203, 317), False, True),
    ("blue64_p32_grad", "blue_noise", {}, ("palr", 32), ("grad", 333, 500), False, True),
    ("blue32_p256_rnd_small", "blue_noise", {"size": 32, "seed": 42}, ("palr", 256), ("rnd", 203, 317, 12), False, True),
    ("bayer4_U16_gamma_grad", "bayer", {"size": "4x4"}, ("U", 16), ("grad", 333, 500), True, True),
    ("bayer8_p256_gamma_grad", "bayer", {"size": "8x8"}, ("palr", 256), ("grad", 333, 500), True, True),
    ("none_p32_gamma_rnd", "none", {}, ("palr", 32), ("rnd", 120, 160, 9), True, True),
    ("ign_p32_gamma_rnd", "IGN", {}, ("palr", 32), ("rnd", 120, 160, 9), True, True),
    ("polka_default_p32_grad", "polka_dot", {}, ("palr", 32), ("grad", 203, 317), False, True),
    ("polka_t5_g07_p256_rnd", "polka_dot", {"tile_size": 5, "gamma": 0.7}, ("palr", 256), ("rnd", 203, 317, 13), False, True),
    ("polka_t32_g3_U16_gamma_grad", "polka_dot", {"tile_size": 32, "gamma": 3.0}, ("U", 16), ("grad", 97, 131), True, True),
    ("perceptual_p16_rnd", "perceptual", {}, ("palr", 16), ("rnd", 40, 56, 6), False, True),
    ("perceptual_U16_grad", "perceptual", {}, ("U", 16), ("grad", 48, 64), False, True),
    ("perceptual_p32_gamma_rnd", "perceptual", {}, ("palr", 32), ("rnd", 33, 47, 7), True, True),
    ("hybrid_default_p16_rnd", "hybrid", {}, ("palr", 16), ("rnd", 40, 56, 6), False, True),
    ("hybrid_l07_c13_U16_grad", "hybrid", {"lum_factor": 0.7, "col_factor": 1.3}, ("U", 16), ("grad", 48, 64), False, True),
    ("adaptive_default_p16_rnd", "adaptive_variance", {}, ("palr", 16), ("rnd", 40, 56, 6), False, True),
    ("adaptive_t50_r2_U16_grad", "adaptive_variance", {"var_threshold": 50.0, "window_radius": 2}, ("U", 16), ("grad", 48, 64), False, True),
    ("adaptive_t900_r1_p32_gamma_rnd", "adaptive_variance", {"var_threshold": 900.0}, ("palr", 32), ("rnd", 33, 47, 7), True, True),
    ("ostro_false_p16_rnd", "ostromoukhov", {"serpentine": "false"}, ("palr", 16), ("rnd", 40, 56, 6), False, True),
    ("ostro_true_U16_grad", "ostromoukhov", {"serpentine": "true"}, ("U", 16), ("grad", 48, 64), False, True),
    ("ostro_false_p64_gamma_grad", "ostromoukhov", {}, ("palr", 64), ("grad", 33, 47), True, True),
    ("ed_fs_p16_rnd", "error_diffusion", {"variant": "floyd_steinberg"}, ("palr", 16), ("rnd", 120, 160, 5), False, True),
    ("ed_default_p256_grad", "error_diffusion", {}, ("palr", 256), ("grad", 64, 96), False, True),
    ("ed_fs_U16_gamma_grad", "error_diffusion", {"variant": "floyd_steinberg"}, ("U", 16), ("grad", 64, 96), True, True),
    ("ed_bogus_U16_grad", "error_diffusion", {"variant": "nope"}, ("U", 16), ("grad", 40, 56), False, True),
    ("ed_jjn_p32_rnd_serp", "error_diffusion", {"variant": "jjn", "serpentine": "true"}, ("palr", 32), ("rnd", 50, 70, 8), False, True),
    ("ed_fs_p16_tiny_w1", "error_diffusion", {"variant": "floyd_steinberg"}, ("palr", 16), ("rnd", 17, 1, 2), False, True),
    ("ed_stucki_p16_tiny_h1", "error_diffusion", {"variant": "stucki"}, ("palr", 16), ("rnd", 1, 23, 2), False, True),
]
# Palettes extracted from the image itself (the reference's default: palette=None => median cut, dithering_lib.py:1960-1966):
# the palette spec ("mc", K) is resolved with the REFERENCE's reduce_colors and recorded as a plain list.
CASES += [
    ("bayer8_mc64_smooth", "bayer", {"size": "8x8"}, ("mc", 64), ("imgl", 240, 320, 5, "smooth"), False, True),
    ("bayer8_mc256_dark", "bayer", {"size": "8x8"}, ("mc", 256), ("imgl", 240, 320, 6, "dark"), False, True),
    ("none_mc16_dark", "none", {}, ("mc", 16), ("imgl", 200, 301, 7, "dark"), False, True),
    ("ign_mc128_smooth", "IGN", {}, ("mc", 128), ("imgl", 200, 301, 8, "smooth"), False, True),
    ("blue32_mc256_smooth", "blue_noise", {"size": 32, "seed": 7}, ("mc", 256), ("imgl", 200, 301, 9, "smooth"), False, True),
    ("bayer8_mc64_gamma_smooth", "bayer", {"size": "8x8"}, ("mc", 64), ("imgl", 120, 160, 10, "smooth"), True, True),
    ("ed_fs_mc16_smooth", "error_diffusion", {"variant": "floyd_steinberg"}, ("mc", 16), ("imgl", 96, 128, 11, "smooth"), False, True),
    ("ed_jjn_mc256_dark", "error_diffusion", {"variant": "jjn"}, ("mc", 256), ("imgl", 64, 96, 12, "dark"), False, True),
]
# BASELINE.json's configurations at their FULL sizes (hashes only): C1 exactly as examples/image_basic.json runs it (Bayer
# default 4x4, 16 colours by median cut of the image itself), one C5 frame, the whole C4 image (the GPU tests dither it
# in 8 row bands), and the C3 frame (pure-Python error diffusion: ~7 minutes in the reference).
CASES += [
    ("c1_bayer4_mc16_rnd512", "bayer", {"size": "4x4"}, ("mc", 16), ("rnd", 512, 512, 1234), False, False),
    ("c5_bayer4_U16_rnd1080", "bayer", {"size": "4x4"}, ("U", 16), ("rnd", 1080, 1920, 0), False, False),
    ("c4_blue64_p32_rnd8k", "blue_noise", {"size": 64, "seed": 42}, ("palr", 32), ("rnd", 4320, 7680, 99), False, False),
    ("c3_ed_fs_U16_rnd4k", "error_diffusion", {"variant": "floyd_steinberg", "serpentine": "false"}, ("U", 16),
     ("rnd", 2160, 3840, 1234), False, False),
]
# Round 5: error diffusion with 17..256 colours (the hierarchical <= 4-entry nearest table of ed_nearest.hip.h and its policies)
# on images of two to four 64-row bands, hashed by the reference: random, median-cut (table kept) and dark median-cut (lists kept)
# palettes, the boundary size 17, use_gamma.
CASES += [
    ("r5_ed_fs_p32_rnd", "error_diffusion", {"variant": "floyd_steinberg"}, ("palr", 32), ("rnd", 150, 200, 51), False, False),
    ("r5_ed_fs_p64_grad", "error_diffusion", {"variant": "floyd_steinberg"}, ("palr", 64), ("grad", 200, 150), False, False),
    ("r5_ed_atkinson_p128_rnd", "error_diffusion", {}, ("palr", 128), ("rnd", 140, 180, 52), False, False),
    ("r5_ed_fs_p256_rnd", "error_diffusion", {"variant": "floyd_steinberg"}, ("palr", 256), ("rnd", 200, 260, 53), False, False),
    ("r5_ed_jjn_p100_grad", "error_diffusion", {"variant": "jjn"}, ("palr", 100), ("grad", 130, 170), False, False),
    ("r5_ed_stucki_p17_rnd", "error_diffusion", {"variant": "stucki"}, ("palr", 17), ("rnd", 100, 140, 54), False, False),
    ("r5_ed_sierra_p200_gamma_grad", "error_diffusion", {"variant": "sierra"}, ("palr", 200), ("grad", 100, 130), True, False),
    ("r5_ed_fs_mc64_smooth", "error_diffusion", {"variant": "floyd_steinberg"}, ("mc", 64), ("imgl", 160, 200, 55, "smooth"), False, False),
    ("r5_ed_atkinson_mc128_smooth", "error_diffusion", {}, ("mc", 128), ("imgl", 150, 190, 56, "smooth"), False, False),
    ("r5_ed_fs_mc256_dark", "error_diffusion", {"variant": "floyd_steinberg"}, ("mc", 256), ("imgl", 140, 180, 57, "dark"), False, False),
    ("r5_ed_burkes_mc32_dark", "error_diffusion", {"variant": "burkes"}, ("mc", 32), ("imgl", 130, 170, 58, "dark"), False, False),
]
# Round 5: the variable-coefficient diffusers (vardiff.hip) over two or three 64-row bands and with more than 16 colours (their
# fixtures so far were single-band, at most 64 colours), hashed by the reference.
CASES += [
    ("r5_perceptual_p64_rnd", "perceptual", {}, ("palr", 64), ("rnd", 150, 130, 61), False, False),
    ("r5_perceptual_mc32_smooth", "perceptual", {}, ("mc", 32), ("imgl", 140, 120, 62, "smooth"), False, False),
    ("r5_hybrid_p32_grad", "hybrid", {}, ("palr", 32), ("grad", 140, 160), False, False),
    ("r5_hybrid_l12_c05_p128_rnd", "hybrid", {"lum_factor": 1.2, "col_factor": 0.5}, ("palr", 128), ("rnd", 130, 110, 63), False, False),
    ("r5_adaptive_p128_rnd", "adaptive_variance", {}, ("palr", 128), ("rnd", 130, 150, 64), False, False),
    ("r5_adaptive_t200_r3_mc64_dark", "adaptive_variance", {"var_threshold": 200.0, "window_radius": 3}, ("mc", 64), ("imgl", 150, 100, 65, "dark"), False, False),
    ("r5_ostro_false_p32_rnd", "ostromoukhov", {"serpentine": "false"}, ("palr", 32), ("rnd", 160, 140, 66), False, False),
    ("r5_ostro_true_p256_grad", "ostromoukhov", {"serpentine": "true"}, ("palr", 256), ("grad", 130, 100), False, False),
    ("r5_ostro_false_U16_gamma_rnd", "ostromoukhov", {"serpentine": "false"}, ("U", 16), ("rnd", 140, 90, 67), True, False),
]
# Round 5: palettes of 257..1024 colours (the 10-bit index keys, the cell table with split nodes beyond LDS), hashed by the reference.
CASES += [
    ("r5_bayer8_p300_rnd", "bayer", {"size": "8x8"}, ("palr", 300), ("rnd", 203, 317, 71), False, False),
    ("r5_none_p1024_grad", "none", {}, ("palr", 1024), ("grad", 240, 320), False, False),
    ("r5_ign_p700_rnd", "IGN", {"scale": 1.7, "seed": 9}, ("palr", 700), ("rnd", 200, 300, 72), False, False),
    ("r5_blue32_p512_gamma_grad", "blue_noise", {"size": 32, "seed": 5}, ("palr", 512), ("grad", 150, 210), True, False),
    ("r5_ed_fs_p300_rnd", "error_diffusion", {"variant": "floyd_steinberg"}, ("palr", 300), ("rnd", 130, 170, 73), False, False),
    ("r5_ed_atkinson_p1024_grad", "error_diffusion", {}, ("palr", 1024), ("grad", 100, 140), False, False),
]
# Round 5, late: the unclamped diffusers above 256 colours (extended lists with ten-bit entries) and under use_gamma with 64 / 256 colours
# (palettes crowded at the dark faces of the cube: the octree below the outermost cells), hashed by the reference.
CASES += [
    ("r5_perceptual_p300_rnd", "perceptual", {}, ("palr", 300), ("rnd", 130, 110, 81), False, False),
    ("r5_hybrid_p1024_grad", "hybrid", {"lum_factor": 0.8, "col_factor": 0.6}, ("palr", 1024), ("grad", 120, 140), False, False),
    ("r5_adaptive_p512_gamma_rnd", "adaptive_variance", {"var_threshold": 120.0}, ("palr", 512), ("rnd", 110, 130, 82), True, False),
    ("r5_perceptual_p256_gamma_grad", "perceptual", {}, ("palr", 256), ("grad", 140, 120), True, False),
    ("r5_hybrid_p64_gamma_rnd", "hybrid", {}, ("palr", 64), ("rnd", 150, 100, 83), True, False),
    ("r5_ostro_false_p700_rnd", "ostromoukhov", {"serpentine": "false"}, ("palr", 700), ("rnd", 140, 100, 84), False, False),
]
# Round 3: the kernels for crowded palettes (ordered_compact_kernel) and for use_gamma (ordered_compact_float_kernel) at 1080p,
# hashed by the reference: image-like content with its own median-cut 256 palette (the reference's default palette source),
# all three ordered decision modes, and the float path on noise and on image-like content.
CASES += [
    ("r3_bayer8_mc256_smooth1080", "bayer", {"size": "8x8"}, ("mc", 256), ("imgl", 1080, 1920, 31, "smooth"), False, False),
    ("r3_none_mc256_dark1080", "none", {}, ("mc", 256), ("imgl", 1080, 1920, 32, "dark"), False, False),
    ("r3_ign_mc128_smooth1080", "IGN", {"scale": 1.5, "seed": 3}, ("mc", 128), ("imgl", 1080, 1920, 33, "smooth"), False, False),
    ("r3_blue64_mc256_dark1080", "blue_noise", {"size": 64, "seed": 42}, ("mc", 256), ("imgl", 1080, 1920, 34, "dark"), False, False),
    ("r3_bayer8_p256_gamma_rnd1080", "bayer", {"size": "8x8"}, ("palr", 256), ("rnd", 1080, 1920, 35), True, False),
    ("r3_none_p256_gamma_rnd1080", "none", {}, ("palr", 256, 9), ("rnd", 1080, 1920, 36), True, False),
    ("r3_ign_mc256_gamma_smooth1080", "IGN", {}, ("mc", 256), ("imgl", 1080, 1920, 37, "smooth"), True, False),
    ("r3_bayer4_p64_gamma_grad1080", "bayer", {"size": "4x4"}, ("palr", 64), ("grad", 1080, 1920), True, False),
]
for v in ED_VARIANTS:
    for s in ("false", "true"):
        CASES.append((f"ed_{v}_{s}_U16_grad", "error_diffusion", {"variant": v, "serpentine": s},
                      ("U", 16), ("grad", 64, 96), False, True))


# More k-means palettes from the reference where it is deterministic (<= 10 000 pixels, dithering_lib.py:1845-1857):
# (name, input spec, K, random_state).  Recorded per case: the reference palette, sklearn's own k-means++ picks for that
# random_state on the mean-centred data (what KMeans.fit seeds with), its centres, inertia and iteration count.
KM_EXTRA = [
    ("kmx_smooth_k4", ("imgl", 64, 96, 21, "smooth"), 4, 42),
    ("kmx_dark_k16", ("imgl", 90, 110, 22, "dark"), 16, 42),
    ("kmx_smooth_k64", ("imgl", 100, 100, 23, "smooth"), 64, 42),
    ("kmx_rnd_k2", ("rnd", 50, 60, 24), 2, 42),
    ("kmx_rnd_k16_rs7", ("rnd", 70, 70, 25), 16, 7),
    ("kmx_grad_k32_rs123", ("grad", 99, 101), 32, 123),
    ("kmx_exact10000_k8", ("rnd", 100, 100, 26), 8, 42),
    ("kmx_fewcolours_k8", ("grad", 16, 20), 8, 42),
]


def append_kmeans_extra(kat, npz):
    from sklearn.cluster import KMeans, kmeans_plusplus
    import video_processor as vp
    done = kat["misc"].setdefault("kmeans_extra", {})
    for nm, ispec, K, rs in KM_EXTRA:
        if nm in done:
            continue
        arr = make_input(ispec)
        pal = dl.ColorReducer.generate_kmeans_palette(Image.fromarray(arr), K, random_state=rs)
        X = arr.reshape(-1, 3).astype(np.float64)
        _, init_idx = kmeans_plusplus(X - X.mean(axis=0), K, random_state=rs)
        km = KMeans(n_clusters=K, random_state=rs).fit(arr.reshape(-1, 3))
        km2 = KMeans(n_clusters=K, init=X[init_idx], n_init=1).fit(arr.reshape(-1, 3))
        assert np.allclose(km.cluster_centers_, km2.cluster_centers_, atol=1e-9), "init replay mismatch"
        assert [tuple(c) for c in km.cluster_centers_.astype(int)] == [tuple(int(v) for v in c) for c in pal]
        npz[f"{nm}_palette"] = np.array(pal, np.int32)
        npz[f"{nm}_init_idx"] = np.asarray(init_idx, np.int32)
        npz[f"{nm}_centers"] = km.cluster_centers_
        done[nm] = dict(input=list(ispec), K=K, random_state=rs, inertia=float(km.inertia_), n_iter=int(km.n_iter_))
        print(nm, km.n_iter_, km.inertia_, flush=True)
    # examples/image_pixelized.json as dither_cli.process_single_image runs it (dither_cli.py:516, 423, 546-566): regular
    # pixelization to max_size 64, k-means 16 (random_state=42) of the pixelized image, error diffusion with its defaults,
    # use_gamma, final NEAREST resize x8 - on a synthetic 400x300 stand-in for the absent test_300.png
    if "pixelized_example" not in kat["misc"]:
        src = Image.fromarray(imgl(300, 400, 27, "smooth"))
        small = vp.pixelize_regular(src, 64)
        pal = dl.ColorReducer.generate_kmeans_palette(small, 16, random_state=42)
        d = dl.ImageDitherer(16, dl.DitherMode("error_diffusion"), [tuple(int(v) for v in c) for c in pal], True, {})
        out = d.apply_dithering(small)
        big = out.resize((out.width * 8, out.height * 8), Image.Resampling.NEAREST)
        kat["misc"]["pixelized_example"] = dict(input=["imgl", 300, 400, 27, "smooth"], max_size=64, K=16, multiplier=8,
                                                small_size=list(small.size), palette=[[int(v) for v in c] for c in pal],
                                                h_small=H(np.array(small)), h_out=H(np.array(out)), h_big=H(np.array(big)))
        npz["pixelized_example_out"] = np.array(out)
        print("pixelized_example", small.size, H(np.array(out)), flush=True)


# Few-colour images, where a cluster's exact mean is an integer (flat-colour / pixel-art images, single-colour clusters,
# K >= distinct colours): sklearn accumulates mean-centred float64 members per thread chunk, adds the partial sums in the
# order the threads finish, divides and adds the mean back -- the centre lands ON the integer or 1 ulp below it, so the
# reference's `astype(int)` yields the colour or colour - 1, and which one differs from RUN TO RUN of the reference (round-3
# advisor; measured here: 7 of 20 such images gave more than one palette in five runs).  Recorded: every distinct palette
# seen in REPEATS runs of the reference, and the centres of the first run.
#   (name, pixels n, distinct colours, K, data seed)
KM_FEW = [("kmf_5col_k5", 1200, 5, 5, 1), ("kmf_8col_k8", 2932, 8, 8, 2), ("kmf_3col_k6", 800, 3, 6, 3),
          ("kmf_6col_k4", 2500, 6, 4, 4), ("kmf_flat_k3", 400, 1, 3, 5)]
KM_FEW_REPEATS = 8


def few_colour_pixels(n, nc, seed):
    rs = np.random.RandomState(seed)
    cols = rs.randint(0, 256, (nc, 3))
    return cols[rs.randint(0, nc, n)].astype(np.uint8)


def record_kmeans_host(kat):
    """Which BLAS / SIMD / thread configuration the k-means fixtures were recorded on (exact ties are decided by its
    rounding order; the kmf_* palettes also by thread scheduling)."""
    import threadpoolctl
    info = threadpoolctl.threadpool_info()
    blas = next((i for i in info if i.get("user_api") == "blas" and "numpy" in i.get("filepath", "")), {})
    omp = next((i for i in info if i.get("user_api") == "openmp"), {})
    simd = np.__config__.show(mode="dicts").get("SIMD Extensions", {}).get("found", [])
    kat["versions"]["kmeans_fixture_host"] = {
        "note": "the k-means fixtures (km*, kmx_*, kmf_*) depend on the host's BLAS / SIMD rounding order where two centres are "
                "exactly equidistant (label_f64 in kmeans.hip restates THIS configuration) and, for kmf_*, on thread scheduling",
        "blas": f"OpenBLAS {blas.get('version')} ({os.path.basename(os.path.dirname(blas.get('filepath', '')))}/libscipy_openblas64_), "
                f"threading {blas.get('threading_layer')}, architecture {blas.get('architecture')} (AVX-512 dgemm micro-kernel, FMA)",
        "numpy_simd_found": " ".join(simd), "openmp": f"libgomp (scikit_learn.libs), {omp.get('num_threads')} threads",
        "cpus": os.cpu_count()}


def append_kmeans_few(kat, npz):
    import warnings
    done = kat["misc"].setdefault("kmeans_few", {})
    for nm, n, nc, K, seed in KM_FEW:
        if nm in done:
            continue
        px = few_colour_pixels(n, nc, seed)
        img = Image.fromarray(px.reshape(-1, 1, 3))   # an n x 1 image: <= 10 000 pixels, no sampling
        pals = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")            # (sklearn warns when there are fewer distinct points than clusters)
            for _ in range(KM_FEW_REPEATS):
                pal = [[int(v) for v in c] for c in dl.ColorReducer.generate_kmeans_palette(img, K, random_state=42)]
                if pal not in pals:
                    pals.append(pal)
        npz[f"{nm}_palettes"] = np.array(pals, np.int32)
        done[nm] = dict(n=n, colours=nc, K=K, seed=seed, repeats=KM_FEW_REPEATS, distinct_reference_palettes=len(pals))
        print(nm, "distinct palettes in", KM_FEW_REPEATS, "runs of the reference:", len(pals), flush=True)


BLUE_EXTRA = [(96, 1), (128, 42), (130, 7)]


def append_new():
    """Adds the cases of CASES that kat.json does not hold yet (and their median-cut palettes) and the KM_EXTRA k-means
    fixtures without touching the rest."""
    with open(os.path.join(HERE, "kat.json")) as f:
        kat = json.load(f)
    npz = dict(np.load(os.path.join(HERE, "small.npz")))
    append_kmeans_extra(kat, npz)
    append_kmeans_few(kat, npz)
    if "kmeans_fixture_host" not in kat["versions"]:
        record_kmeans_host(kat)
    # round 5: blue-noise matrices at the largest size the GUI offers (128: the last one whose arrays fit the kernel's LDS), beyond
    # it (130: the global-memory instance) and in between -- the reference needs n^2 Python steps (2-3 minutes each)
    for size, seed in BLUE_EXTRA:
        key = f"blue_{size}_{seed}"
        if key in kat["misc"]:
            continue
        bn = dl.generate_blue_noise(size, seed)
        npz[key] = bn
        kat["misc"][key] = H(bn)
        print(key, H(bn), flush=True)
    have = {c["name"] for c in kat["cases"]}
    for name, mode, params, pspec, ispec, gamma, keep in CASES:
        if name in have:
            continue
        arr = make_input(ispec)
        if pspec[0] == "mc":
            pal = [tuple(int(v) for v in c) for c in dl.ColorReducer.reduce_colors(Image.fromarray(arr), pspec[1])]
            kat["misc"]["median_cut"][f"{ispec[0]}:{':'.join(str(v) for v in ispec[1:])}_{pspec[1]}"] = [list(c) for c in pal]
            pspec = ("list", [list(c) for c in pal])
        else:
            pal = make_palette(pspec)
        out = run_ref(arr, pal, mode, params, gamma)
        kat["cases"].append(dict(name=name, mode=mode, params=params, palette=list(pspec), input=list(ispec),
                                 gamma=gamma, h_in=H(arr), h_out=H(out), full=keep))
        if keep:
            npz["out_" + name] = out
        print(name, H(arr), H(out), flush=True)
    np.savez_compressed(os.path.join(HERE, "small.npz"), **npz)
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)
    print("now", len(npz), "arrays,", len(kat["cases"]), "cases")


NUMBA_PROBE_PALETTE = [[84.0, 50.0 + 2.0 ** -18, 50.0], [116.0, 50.0, 50.0]]   # tests/test_oracle_golden.py: _numba_reading_probe
NUMBA_HYBRID = [(1.0, 0.2), (1.4, 0.0), (0.3, 1.0)]


def record_numba():
    """The reference's numba branches -- _error_diffusion_numba (dithering_lib.py:213-308, dispatched at :638-653) and
    _hybrid_numba (:1396-1494, dispatched at :1114-1125) -- run by the reference itself, on a host that has numba.  Both
    dispatches sit in try/except and fall back to the pure-Python loop silently, so the two jitted functions are wrapped and
    every case checks that the numba function was entered and returned.  Writes tests/golden/numba.json (hashes, versions,
    case list) and numba.npz (every output in full: the images are small); tests/test_oracle_golden.py and the GPU tests
    compare with them when they exist and report 'unpinned' when they do not.  Nothing else in tests/golden is touched."""
    try:
        import numba
    except Exception as e:  # noqa: BLE001
        print(f"numba is not importable here ({type(e).__name__}: {e}): nothing recorded; SURVEY row a7 stays parity-unpinned.\n"
              "Run this command on a host with numba + the reference checkout (DITHER_PIE_REFERENCE=/path/to/dither_pie).")
        return 2
    if not getattr(dl, "_NUMBA_AVAILABLE", False):
        print("the reference did not enable its numba branch (_NUMBA_AVAILABLE is False): nothing recorded")
        return 2
    entered = {"ed": 0, "hy": 0}
    orig_ed, orig_hy = dl._error_diffusion_numba, dl._hybrid_numba

    def ed(*a, **k):
        r = orig_ed(*a, **k)
        entered["ed"] += 1
        return r

    def hy(*a, **k):
        r = orig_hy(*a, **k)
        entered["hy"] += 1
        return r

    dl._error_diffusion_numba, dl._hybrid_numba = ed, hy
    import scipy
    rec = {"versions": {"numba": numba.__version__, "numpy": np.__version__, "scipy": scipy.__version__,
                        "python": sys.version.split()[0]}, "cases": []}
    npz = {}

    def through(kind, before):
        assert entered[kind] == before + 1, f"the reference fell back to its pure-Python loop ({kind}): numba failed to compile?"

    # (1) the 8 kernels x serpentine off / on through the public entry point, two inputs (one with use_gamma: float palette)
    inputs = [(("rnd", 48, 64, 11), ("palr", 16, 3), False), (("grad", 40, 56), ("U", 16), True)]
    for ispec, pspec, gamma in inputs:
        arr, pal = make_input(ispec), make_palette(pspec)
        for variant in ED_VARIANTS:
            for serp in ("false", "true"):
                n0 = entered["ed"]
                out = run_ref(arr, pal, "error_diffusion", {"variant": variant, "serpentine": serp}, gamma)
                through("ed", n0)
                name = f"nb_ed_{variant}_{serp}_{ispec[0]}{'_gamma' if gamma else ''}"
                rec["cases"].append(dict(name=name, kind="error_diffusion", params={"variant": variant, "serpentine": serp},
                                         palette=list(pspec), input=list(ispec), gamma=gamma, h_in=H(arr), h_out=H(out)))
                npz[name] = out
                print(name, H(out), flush=True)
    # (2) the input on which the float32 and the float64 reading of the scan choose differently, and the strip of tiny values
    # against a far colour (float64 error), straight through the strategy contract dither(pixels, palette_arr, (h, w))
    probe_pal = np.array(NUMBA_PROBE_PALETTE, np.float32)
    st = dl.ErrorDiffusionDitherStrategy(variant="floyd_steinberg", serpentine="false")
    n0 = entered["ed"]
    res = st.dither(np.array([[100.0, 50.0, 50.0]], np.float32), probe_pal, (1, 1))
    through("ed", n0)
    npz["nb_probe_rows"] = np.asarray(res, np.float32)
    rec["probe"] = {"pixels": [[100, 50, 50]], "palette_f32": NUMBA_PROBE_PALETTE, "chosen_row": np.asarray(res).reshape(-1).tolist()}
    pal2 = np.array([[255.0, 255.0, 255.0], [0.3, 0.3, 0.3], [17.7, 200.1, 3.3]], np.float32)
    strip = rnd(3, 40, 8)
    for serp in ("false", "true"):
        n0 = entered["ed"]
        res = dl.ErrorDiffusionDitherStrategy(variant="jjn", serpentine=serp).dither(
            strip.reshape(-1, 3).astype(np.float32), pal2, (3, 40))
        through("ed", n0)
        npz[f"nb_strip_jjn_{serp}"] = np.asarray(res, np.float32).reshape(3, 40, 3)
    rec["strip"] = {"input": ["rnd", 3, 40, 8], "palette_f32": pal2.tolist(), "variant": "jjn"}
    # (3) _hybrid_numba, three settings, through the public entry point
    for (lum, col), (ispec, pspec, gamma) in zip(NUMBA_HYBRID, [(("rnd", 13, 17, 3), ("U", 16), False),
                                                               (("grad", 9, 21), ("palr", 9, 5), True),
                                                               (("rnd", 11, 8, 4), ("palr", 5, 2), False)]):
        arr, pal = make_input(ispec), make_palette(pspec)
        n0 = entered["hy"]
        out = run_ref(arr, pal, "hybrid", {"lum_factor": lum, "col_factor": col}, gamma)
        through("hy", n0)
        name = f"nb_hybrid_{lum}_{col}"
        rec["cases"].append(dict(name=name, kind="hybrid", params={"lum_factor": lum, "col_factor": col}, palette=list(pspec),
                                 input=list(ispec), gamma=gamma, h_in=H(arr), h_out=H(out)))
        npz[name] = out
        print(name, H(out), flush=True)
    np.savez_compressed(os.path.join(HERE, "numba.npz"), **npz)
    with open(os.path.join(HERE, "numba.json"), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print("wrote numba.json / numba.npz:", len(rec["cases"]), "cases; commit both -- row a7 is then pinned by the reference")
    return 0


def main():
    import scipy
    import sklearn
    import PIL
    from scipy.spatial import cKDTree
    from sklearn.cluster import KMeans, kmeans_plusplus

    kat = {"versions": {"numpy": np.__version__, "scipy": scipy.__version__, "sklearn": sklearn.__version__,
                        "pillow": PIL.__version__, "python": sys.version.split()[0]},
           "cases": [], "tables": {}, "misc": {}}
    npz = {}

    # ---- threshold tables
    for name in ["BAYER2x2", "BAYER4x4", "BAYER8x8", "BAYER16x16", "PSX4x4"]:
        m = getattr(dl.DitherUtils, name)
        kat["tables"][name] = H(m)
        npz["table_" + name] = m

    # ---- gamma LUTs (dithering_lib.py:1957-1959, 1986-1989) and the palette linearisation table
    k = np.arange(256, dtype=np.uint8)
    lut_in = np.clip(dl.DitherUtils.srgb_to_linear(k.astype(np.float32) / 255.0) * 255.0, 0, 255).astype(np.uint8)
    lut_out = np.clip(dl.DitherUtils.linear_to_srgb(np.clip(k.astype(np.float32) / 255.0, 0, 1)) * 255.0, 0, 255).astype(np.uint8)
    pal_lin = np.clip(dl.DitherUtils.srgb_to_linear(k.astype(np.float32) / 255.0) * 255.0, 0, 255).astype(np.float32)
    npz["lut_in"], npz["lut_out"], npz["pal_lin_table"] = lut_in, lut_out, pal_lin
    kat["misc"]["lut_in"], kat["misc"]["lut_out"], kat["misc"]["pal_lin_table"] = H(lut_in), H(lut_out), H(pal_lin)

    # ---- IGN thresholds
    s = dl.InterleavedGradientNoiseDitherStrategy(1.0, 0)
    npz["ign_8x8_s1_seed0"] = s._generate_thresholds((8, 8))
    kat["misc"]["ign_8x8_s1_seed0"] = H(npz["ign_8x8_s1_seed0"])
    s = dl.InterleavedGradientNoiseDitherStrategy(2.5, 17)
    kat["misc"]["ign_1080_s25_seed17"] = H(s._generate_thresholds((1080, 1920)))
    npz["ign_37x53_s25_seed17"] = s._generate_thresholds((37, 53))
    s = dl.InterleavedGradientNoiseDitherStrategy(0.1, 9999)
    npz["ign_64x64_s01_seed9999"] = s._generate_thresholds((64, 64))
    s = dl.InterleavedGradientNoiseDitherStrategy(10.0, 4321)
    kat["misc"]["ign_4k_s10_seed4321"] = H(s._generate_thresholds((2160, 3840)))

    # ---- blue noise
    for size, seed in [(32, 42), (64, 42), (32, 0), (33, 9999)]:
        bn = dl.generate_blue_noise(size, seed)
        npz[f"blue_{size}_{seed}"] = bn
        kat["misc"][f"blue_{size}_{seed}"] = H(bn)
        print("blue", size, seed, H(bn), flush=True)

    # ---- dither cases
    for name, mode, params, pspec, ispec, gamma, keep in CASES:
        arr = make_input(ispec)
        if pspec[0] == "mc":  # resolved by the reference, recorded as a list
            pal = [tuple(int(v) for v in c) for c in dl.ColorReducer.reduce_colors(Image.fromarray(arr), pspec[1])]
            kat["misc"].setdefault("median_cut_cases", {})[f"{ispec[0]}:{':'.join(str(v) for v in ispec[1:])}_{pspec[1]}"] = [list(c) for c in pal]
            pspec = ("list", [list(c) for c in pal])
        else:
            pal = make_palette(pspec)
        out = run_ref(arr, pal, mode, params, gamma)
        kat["cases"].append(dict(name=name, mode=mode, params=params, palette=list(pspec), input=list(ispec),
                                 gamma=gamma, h_in=H(arr), h_out=H(out), full=keep))
        if keep:
            npz["out_" + name] = out
        print(name, H(arr), H(out), flush=True)

    # ---- cKDTree structures + tie-heavy queries (scipy.spatial.KDTree defaults: leafsize 10)
    tree_pals = {
        "p256": np.array(palr(256), np.float64), "p100": np.array(palr(100, 3), np.float64),
        "p32": np.array(palr(32), np.float64), "p16": np.array(palr(16), np.float64),
        "p11": np.array(palr(11, 2), np.float64),
        "U16": np.array(dl.ColorReducer.generate_uniform_palette(16), np.float64),
        "U64": np.array(dl.ColorReducer.generate_uniform_palette(64), np.float64),
        "U256": np.array(dl.ColorReducer.generate_uniform_palette(256), np.float64),
        "dup40": np.array(palr(20, 4) + palr(20, 4), np.float64),
        "flat": np.array([(i % 7 * 30, 5, 200) for i in range(64)], np.float64),
        "lin256": np.array(pal_lin[np.array(palr(256))], np.float64),
    }
    rs = np.random.RandomState(123)
    for nm, P in tree_pals.items():
        t = cKDTree(P, leafsize=10)
        nodes = []

        def walk(n):
            i = len(nodes)
            nodes.append(None)
            if n.split_dim == -1:
                nodes[i] = (-1, 0.0, n.start_idx, n.end_idx, -1, -1)
            else:
                l = walk(n.lesser)
                g = walk(n.greater)
                nodes[i] = (n.split_dim, n.split, n.start_idx, n.end_idx, l, g)
            return i

        walk(t.tree)
        npz[f"tree_{nm}_pts"] = P
        npz[f"tree_{nm}_indices"] = np.asarray(t.indices, np.int32)
        npz[f"tree_{nm}_nodes"] = np.array([(a, c, d, e, f) for a, b, c, d, e, f in nodes], np.int32)
        npz[f"tree_{nm}_splits"] = np.array([b for a, b, c, d, e, f in nodes], np.float64)
        # queries: random integer points + points built to tie (midpoints / mirrored points)
        q = [rs.randint(0, 256, (4000, 3)).astype(np.float64)]
        a = P[rs.randint(0, len(P), 3000)]
        b = P[rs.randint(0, len(P), 3000)]
        q.append(np.floor((a + b) / 2))
        q.append(P[rs.randint(0, len(P), 500)])
        q.append(np.clip(P[rs.randint(0, len(P), 1500)] + rs.randint(-3, 4, (1500, 3)), 0, 255))
        q = np.concatenate(q).astype(np.float32).astype(np.float64)  # what travels is f32
        d1, i1 = t.query(q, k=1)
        d2, i2 = t.query(q, k=2)
        npz[f"tree_{nm}_q"] = q.astype(np.float32)
        npz[f"tree_{nm}_i1"] = i1.astype(np.int32)
        npz[f"tree_{nm}_i2"] = i2.astype(np.int32)
        npz[f"tree_{nm}_d2"] = d2
    kat["misc"]["tree_palettes"] = list(tree_pals.keys())

    # ---- k-means (dithering_lib.py:1845-1857): <= 10000 px => deterministic in the reference
    for nm, (hh, ww, seed, K) in {"km8": (100, 100, 31, 8), "km16": (80, 125, 32, 16), "km32": (100, 100, 33, 32)}.items():
        arr = rnd(hh, ww, seed) if nm != "km16" else grad(hh, ww)
        pal = dl.ColorReducer.generate_kmeans_palette(Image.fromarray(arr), K)
        X = arr.reshape(-1, 3).astype(np.float64)
        Xc = X - X.mean(axis=0)
        _, init_idx = kmeans_plusplus(Xc, K, random_state=42)
        km = KMeans(n_clusters=K, random_state=42).fit(arr.reshape(-1, 3))
        km2 = KMeans(n_clusters=K, init=X[init_idx], n_init=1).fit(arr.reshape(-1, 3))
        assert np.allclose(km.cluster_centers_, km2.cluster_centers_, atol=1e-9), "init replay mismatch"
        npz[f"{nm}_palette"] = np.array(pal, np.int32)
        npz[f"{nm}_init_idx"] = np.asarray(init_idx, np.int32)
        npz[f"{nm}_centers"] = km.cluster_centers_
        kat["misc"][nm] = dict(h=hh, w=ww, seed=seed, K=K, kind="grad" if nm == "km16" else "rnd",
                               inertia=float(km.inertia_), n_iter=int(km.n_iter_))
        print(nm, km.n_iter_, km.inertia_, flush=True)

    # ---- palette producers used by the fixtures / benches
    for n in (2, 8, 16, 27, 64, 256):
        npz[f"uniform_{n}"] = np.array(dl.ColorReducer.generate_uniform_palette(n), np.int32)

    # ---- median cut (ColorReducer.reduce_colors, dithering_lib.py:1813-1843) incl. the palette=None paths
    mc = {}
    for nm, arr in {"rnd40x50": rnd(40, 50, 41), "grad64x96": grad(64, 96)}.items():
        for n in (1, 2, 8, 16, 20):
            mc[f"{nm}_{n}"] = [list(map(int, c)) for c in dl.ColorReducer.reduce_colors(Image.fromarray(arr), n)]
    mc.update(kat["misc"].pop("median_cut_cases", {}))
    kat["misc"]["median_cut"] = mc
    for gamma in (False, True):
        d = dl.ImageDitherer(8, dl.DitherMode.BAYER, None, gamma, {"size": "4x4"})
        arr = rnd(48, 64, 17)
        out = np.array(d.apply_dithering(Image.fromarray(arr)))
        kat["misc"][f"auto_palette_gamma{int(gamma)}"] = dict(palette=[list(map(int, c)) for c in d.palette], h_out=H(out))

    # ---- video helpers (video_processor.py:547-560, 408-415)
    import video_processor as vp
    kat["misc"]["even_dims"] = [[a, b, c, list(vp.NeuralPixelizer._compute_even_dimensions(a, b, c))]
                                for a, b, c in [(1920, 1080, 64), (1080, 1920, 65), (300, 300, 33), (641, 359, 64),
                                                (100, 37, 17), (37, 100, 128)]]
    img = Image.fromarray(rnd(37, 53, 5))
    npz["pixelize_regular_37x53_to16"] = np.array(vp.pixelize_regular(img, 16))
    npz["final_resize_37x53_x3"] = np.array(vp._apply_final_resize_to_frame(img, 3))

    npz["ostro_table"] = np.array(dl.OstromoukhovDitherStrategy.COEFFS_TABLE, np.int32)
    a = rnd(37, 53, 8).astype(np.float32)
    g = 0.299 * a[:, :, 0] + 0.587 * a[:, :, 1] + 0.114 * a[:, :, 2]
    for rad in (1, 2, 5):
        npz[f"varmap_r{rad}"] = dl.AdaptiveVarianceDitherStrategy(300.0, rad)._compute_variance_map(g)
    npz["polka_8_15"] = dl.PolkaDotDitherStrategy(8, 1.5).threshold_matrix
    npz["polka_5_07"] = dl.PolkaDotDitherStrategy(5, 0.7).threshold_matrix

    # ---- strategy parameter metadata (drop-in boundary)
    meta = {}
    for m in dl.DitherMode:
        info = dl.ImageDitherer.get_mode_parameters(m)
        meta[m.value] = info
    kat["misc"]["mode_parameters"] = meta
    kat["misc"]["dither_modes"] = {m.name: m.value for m in dl.DitherMode}

    np.savez_compressed(os.path.join(HERE, "small.npz"), **npz)
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)
    print("wrote", len(npz), "arrays,", len(kat["cases"]), "cases")


if __name__ == "__main__":
    if "--numba" in sys.argv:
        sys.exit(record_numba())
    elif "--append" in sys.argv:
        append_new()
    else:
        main()
