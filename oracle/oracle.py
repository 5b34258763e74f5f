"""CPU oracle for dither_pie's hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and there only as the checker / the timed CPU baseline.  The product package
(dither_pie_amd/) never imports it.

Parity status: pinned by tests/golden/ (outputs of the reference itself, generated in the
build container by tests/golden/make_golden.py).  The reference has no tests of its own.

The heavy loops live in dp_oracle.c (plain C, gcc); this file is the numpy part:
  * the uint8 / gamma wrapper of ImageDitherer.apply_dithering   dithering_lib.py:1952-1992
  * threshold tables (as integer numerators)                     dithering_lib.py:1705-1768
  * error-diffusion tap tables                                   dithering_lib.py:107-188
  * strategy parameter defaults                                  dithering_lib.py:408-419, 460-480,
                                                                 508-529, 592-610
  * sklearn KMeans (k-means++ / Lloyd) restated in numpy         dithering_lib.py:1845-1857
    (third-party: scikit-learn, unpinned by the reference, 1.7.2 in the build container)
  * ColorReducer.generate_uniform_palette                        dithering_lib.py:1859-1872
  * the seeded input generators of SURVEY.md Appendix B
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libdp_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_tree_sizeof.restype = C.c_size_t
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_threads.restype = C.c_int
        L.orc_tree_build.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_tree_export.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        L.orc_tree_query_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_ign_thresholds.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]
        L.orc_ordered_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int]
        L.orc_error_diffusion_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int]
        L.orc_error_diffusion_numba_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_double, C.c_int, C.c_int]
        L.orc_hybrid_numba_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_double, C.c_double]
        L.orc_blue_noise.argtypes = [C.c_int, C.c_uint32, C.c_void_p]
        L.orc_var_diffusion_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_uniform_filter1d_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_long, C.c_int]
        L.orc_kmeans_step.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p]
        L.orc_kmeans_step_sk.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p]
        _LIB = L
    return _LIB


def set_threads(n):
    """OpenMP threads used by the row-parallel C loops; returns the number in effect."""
    return lib().orc_set_threads(int(n))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ----------------------------------------------------------------------------- KD-tree
class Tree:
    """scipy.spatial.KDTree(points) (leafsize 10) restated; see dp_oracle.c."""

    def __init__(self, points):
        pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 3))
        self.K = pts.shape[0]
        self._buf = C.create_string_buffer(lib().orc_tree_sizeof())
        rc = lib().orc_tree_build(self._buf, _p(pts), self.K)
        if rc != 0:
            raise ValueError("oracle tree: K out of range")

    def export(self):
        n = 2 * self.K + 2
        idx = np.zeros(self.K, np.int32)
        sd = np.zeros(n, np.int32)
        sp = np.zeros(n, np.float64)
        st, en, le, gr = (np.zeros(n, np.int32) for _ in range(4))
        nn = lib().orc_tree_export(self._buf, _p(idx), _p(sd), _p(sp), _p(st), _p(en), _p(le), _p(gr))
        return dict(indices=idx, split_dim=sd[:nn], split=sp[:nn], start=st[:nn], end=en[:nn],
                    less=le[:nn], greater=gr[:nn])

    def query(self, x, k):
        """returns (squared distances [n,k] f64, indices [n,k] int32)"""
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, 3))
        n = x.shape[0]
        d2 = np.zeros((n, k), np.float64)
        ii = np.zeros((n, k), np.int32)
        lib().orc_tree_query_batch(self._buf, _p(x), n, k, _p(d2), _p(ii))
        return d2, ii


# ----------------------------------------------------------------------------- tables
_BAYER_NUM = {
    "2x2": (4, [[1, 3], [4, 2]]),
    "4x4": (32, [[1, 17, 5, 21], [25, 9, 29, 13], [7, 23, 3, 19], [31, 15, 27, 11]]),
    "8x8": (64, [[1, 33, 9, 41, 3, 35, 11, 43], [49, 17, 57, 25, 51, 19, 59, 27],
                 [13, 45, 5, 37, 15, 47, 7, 39], [61, 29, 53, 21, 63, 31, 54, 22],
                 [4, 36, 12, 44, 2, 34, 10, 42], [52, 20, 60, 28, 50, 18, 58, 26],
                 [16, 48, 8, 40, 14, 46, 6, 38], [64, 32, 56, 24, 62, 30, 54, 22]]),
    "psx4x4": (16, [[1, 9, 3, 11], [13, 5, 15, 7], [3, 11, 1, 9], [15, 7, 13, 5]]),
}


def _bayer16_num():
    # dithering_lib.py:1728-1761; rows 0-7 are the canonical 16x16 Bayer matrix (v+1),
    # rows 8-15 repeat the 8x8 table (x4) on the left half and a shifted copy on the right.
    rows = [
        [1, 129, 33, 161, 9, 137, 41, 169, 3, 131, 35, 163, 11, 139, 43, 171],
        [193, 65, 225, 97, 201, 73, 233, 105, 195, 67, 227, 99, 203, 75, 235, 107],
        [49, 177, 17, 145, 57, 185, 25, 153, 51, 179, 19, 147, 59, 187, 27, 155],
        [241, 113, 209, 81, 249, 121, 217, 89, 243, 115, 211, 83, 251, 123, 219, 91],
        [13, 141, 45, 173, 5, 133, 37, 165, 15, 143, 47, 175, 7, 135, 39, 167],
        [205, 77, 237, 109, 197, 69, 229, 101, 207, 79, 239, 111, 199, 71, 231, 103],
        [61, 189, 29, 157, 53, 181, 21, 149, 63, 191, 31, 159, 55, 183, 23, 151],
        [253, 125, 221, 93, 245, 117, 213, 85, 255, 127, 223, 95, 247, 119, 215, 87],
        [4, 132, 36, 164, 12, 140, 44, 172, 2, 130, 34, 162, 10, 138, 42, 170],
        [196, 68, 228, 100, 204, 76, 236, 108, 194, 66, 226, 98, 202, 74, 234, 106],
        [52, 180, 20, 148, 60, 188, 28, 156, 50, 178, 18, 146, 58, 186, 26, 154],
        [244, 116, 212, 84, 252, 124, 220, 92, 242, 114, 210, 82, 250, 122, 218, 90],
        [16, 144, 48, 176, 8, 136, 40, 168, 14, 142, 46, 174, 6, 134, 38, 166],
        [208, 80, 240, 112, 200, 72, 232, 104, 206, 78, 238, 110, 198, 70, 230, 102],
        [64, 192, 32, 160, 56, 184, 24, 152, 62, 190, 30, 158, 54, 182, 22, 150],
        [256, 128, 224, 96, 248, 120, 216, 88, 254, 126, 222, 94, 246, 118, 214, 86],
    ]
    return 256, rows


_BAYER_NUM["16x16"] = _bayer16_num()


def bayer_matrix(size="4x4"):
    """dithering_lib.py:431-442: unknown size -> 4x4; 'psx' aliases 'psx4x4'."""
    if size == "psx":
        size = "psx4x4"
    den, num = _BAYER_NUM.get(size, _BAYER_NUM["4x4"])
    return (np.array(num, dtype=np.float64) / den).astype(np.float32)


ED_KERNELS = {  # (dx, dy, weight) in list order, divisor -- dithering_lib.py:107-188
    "floyd_steinberg": ([(1, 0, 7), (-1, 1, 3), (0, 1, 5), (1, 1, 1)], 16),
    "jjn": ([(1, 0, 7), (2, 0, 5), (-2, 1, 3), (-1, 1, 5), (0, 1, 7), (1, 1, 5), (2, 1, 3),
             (-2, 2, 1), (-1, 2, 3), (0, 2, 5), (1, 2, 3), (2, 2, 1)], 48),
    "stucki": ([(1, 0, 8), (2, 0, 4), (-2, 1, 2), (-1, 1, 4), (0, 1, 8), (1, 1, 4), (2, 1, 2),
                (-2, 2, 1), (-1, 2, 2), (0, 2, 4), (1, 2, 2), (2, 2, 1)], 42),
    "burkes": ([(1, 0, 8), (2, 0, 4), (-2, 1, 2), (-1, 1, 4), (0, 1, 8), (1, 1, 4), (2, 1, 2)], 32),
    "atkinson": ([(1, 0, 1), (2, 0, 1), (-1, 1, 1), (0, 1, 1), (1, 1, 1), (0, 2, 1)], 8),
    "sierra": ([(1, 0, 5), (2, 0, 3), (-2, 1, 2), (-1, 1, 4), (0, 1, 5), (1, 1, 4), (2, 1, 2),
                (-1, 2, 2), (0, 2, 3), (1, 2, 2)], 32),
    "sierra_two_row": ([(1, 0, 4), (2, 0, 3), (-2, 1, 1), (-1, 1, 2), (0, 1, 3), (1, 1, 2), (2, 1, 1)], 16),
    "sierra_lite": ([(1, 0, 2), (-1, 1, 1), (0, 1, 1)], 4),
}


def ed_kernel(variant):
    """dithering_lib.py:203: unknown variant -> floyd_steinberg."""
    return ED_KERNELS.get(variant, ED_KERNELS["floyd_steinberg"])


# ----------------------------------------------------------------------------- gamma
def srgb_to_linear(c):
    """dithering_lib.py:1788-1794 (float32 arithmetic, numpy power)."""
    c = np.asarray(c, dtype=np.float32)
    low = c <= 0.04045
    out = np.empty_like(c, dtype=np.float32)
    out[low] = c[low] / 12.92
    out[~low] = ((c[~low] + 0.055) / 1.055) ** 2.4
    return out


def linear_to_srgb(c):
    """dithering_lib.py:1796-1802."""
    c = np.asarray(c, dtype=np.float32)
    low = c <= 0.0031308
    out = np.empty_like(c, dtype=np.float32)
    out[low] = c[low] * 12.92
    out[~low] = 1.055 * (c[~low] ** (1.0 / 2.4)) - 0.055
    return out


def gamma_luts():
    """(lut_in, lut_out): the uint8->uint8 maps of dithering_lib.py:1957-1959 and :1986-1989."""
    k = np.arange(256, dtype=np.uint8)
    a01 = k.astype(np.float32) / 255.0
    lut_in = np.clip(srgb_to_linear(a01) * 255.0, 0, 255).astype(np.uint8)
    o01 = k.astype(np.float32) / 255.0
    lut_out = np.clip(linear_to_srgb(np.clip(o01, 0, 1)) * 255.0, 0, 255).astype(np.uint8)
    return lut_in, lut_out


def prepare_palette(palette, use_gamma):
    """-> (pal_f32 [K,3] as seen by the KD-tree, out_colors uint8 [K,3], lut_in or None)

    dithering_lib.py:1970-1974 (palette), :1984-1990 (what a chosen entry becomes)."""
    pal = np.array(palette, dtype=np.float32).reshape(-1, 3)
    if not use_gamma:
        return np.ascontiguousarray(pal), np.ascontiguousarray(pal.astype(np.uint8)), None
    lut_in, lut_out = gamma_luts()
    pal_lin = np.clip(srgb_to_linear(pal / 255.0) * 255.0, 0, 255).astype(np.float32)
    out_lin8 = pal_lin.astype(np.uint8)
    return np.ascontiguousarray(pal_lin), np.ascontiguousarray(lut_out[out_lin8]), lut_in


# ----------------------------------------------------------------------------- thresholds
def ign_thresholds(h, w, scale=1.0, seed=0, y0=0, x0=0):
    out = np.zeros((h, w), np.float32)
    lib().orc_ign_thresholds(h, w, y0, x0, float(scale), int(seed), _p(out))
    return out


_BN_CACHE = {}


def blue_noise(size=64, seed=42):
    key = (int(size), int(seed))
    if key not in _BN_CACHE:
        out = np.zeros((size, size), np.float32)
        if lib().orc_blue_noise(int(size), int(seed), _p(out)) != 0:
            raise MemoryError
        _BN_CACHE[key] = out
    return _BN_CACHE[key]


# ----------------------------------------------------------------------------- dither
MODE_DEFAULTS = {
    "perceptual": {},
    "hybrid": {"lum_factor": 1.0, "col_factor": 0.2},
    "adaptive_variance": {"var_threshold": 300.0, "window_radius": 1},
    "ostromoukhov": {"serpentine": "false"},
    "polka_dot": {"tile_size": 8, "gamma": 1.5},
    "bayer": {"size": "4x4"},
    "blue_noise": {"size": 64, "seed": 42},
    "IGN": {"scale": 1.0, "seed": 0},
    "error_diffusion": {"variant": "atkinson", "serpentine": "false"},
    "none": {},
}


def polka_dot_matrix(tile_size=8, gamma=1.5):
    """dithering_lib.py:733-743"""
    x = np.arange(tile_size)
    xv, yv = np.meshgrid(x, x)
    c = (tile_size - 1) / 2
    dist = np.sqrt((xv - c) ** 2 + (yv - c) ** 2)
    norm = dist / (np.sqrt(c ** 2 + c ** 2) + 1e-9)
    return np.clip(1.0 - (norm ** gamma), 0, 1).astype(np.float32)


def ordered_u8(arr, pal_f32, out_colors, lut_in, mode, thr=None, scale=1.0, seed=0, y0=0, x0=0,
               want_idx=False):
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    h, w, _ = arr.shape
    out = np.empty_like(arr)
    idx = np.empty((h, w), np.int32) if want_idx else None
    K = pal_f32.shape[0]
    m = {"none": 0, "matrix": 1, "ign": 2}[mode]
    th_h = th_w = 1
    if thr is not None:
        thr = np.ascontiguousarray(thr, dtype=np.float32)
        th_h, th_w = thr.shape
    rc = lib().orc_ordered_u8(_p(arr), _p(out), _p(idx), h, w, y0, x0, _p(pal_f32), K, _p(out_colors),
                              _p(lut_in), m, _p(thr), th_h, th_w, float(scale), int(seed))
    if rc != 0:
        raise ValueError("oracle ordered_u8 failed")
    return (out, idx) if want_idx else out


def error_diffusion_u8(arr, pal_f32, out_colors, lut_in, variant="atkinson", serpentine=False):
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    h, w, _ = arr.shape
    out = np.empty_like(arr)
    taps, div = ed_kernel(variant)
    dx = np.array([t[0] for t in taps], np.int32)
    dy = np.array([t[1] for t in taps], np.int32)
    wq = np.array([t[2] / div for t in taps], np.float64)
    rc = lib().orc_error_diffusion_u8(_p(arr), _p(out), h, w, _p(pal_f32), pal_f32.shape[0], _p(out_colors),
                                      _p(lut_in), _p(dx), _p(dy), _p(wq), len(taps), 1 if serpentine else 0)
    if rc != 0:
        raise ValueError("oracle error_diffusion_u8 failed")
    return out


def error_diffusion_numba_u8(arr, pal_f32, out_colors, lut_in, variant="atkinson", serpentine=False):
    """The reference's numba branch (_error_diffusion_numba, dithering_lib.py:213-308) restated in C, typed per numba's
    unification rule: r / g / b are assigned a float32 element AND float64 literals (:239-251), hence float64, and so are
    the distances of the scan and the error that is pushed.  PARITY UNPINNED (fixtures pending): numba cannot be installed
    here, so nothing the reference produced pins it; tests/test_oracle_golden.py checks it against an independent numpy
    transcription of the same lines."""
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    h, w, _ = arr.shape
    out = np.empty_like(arr)
    taps, div = ed_kernel(variant)
    dx = np.array([t[0] for t in taps], np.int32)
    dy = np.array([t[1] for t in taps], np.int32)
    wts = np.array([t[2] for t in taps], np.float32)
    rc = lib().orc_error_diffusion_numba_u8(_p(arr), _p(out), h, w, _p(pal_f32), pal_f32.shape[0], _p(out_colors),
                                            _p(lut_in), _p(dx), _p(dy), _p(wts), float(div), len(taps), 1 if serpentine else 0)
    if rc != 0:
        raise ValueError("oracle error_diffusion_numba_u8 failed")
    return out


def error_diffusion_numba_numpy(arr, pal_f32, out_colors, lut_in, variant="atkinson", serpentine=False, scan="float64"):
    """A second, independent statement of the same lines with numpy scalar types doing the arithmetic (np.float32 /
    np.float64 objects: every operation rounds as the dtype says), each variable carrying the type numba's unification
    gives it (module docstring of dp_oracle.c's restatement).  Slow: small images only.
    scan="float32": the reading of rounds 1-3 (scan and error in float32), kept ONLY so that a test can show an input on
    which the two readings choose differently; nothing else uses it."""
    arr = np.asarray(arr, np.uint8)
    h, w, _ = arr.shape
    taps, div = ed_kernel(variant)
    work = (lut_in[arr] if lut_in is not None else arr).astype(np.float32)
    pal = np.asarray(pal_f32, np.float32)
    weights = np.array([t[2] for t in taps], np.float32)
    divisor = float(div)
    pick = np.zeros((h, w), np.int64)
    T = np.float64 if scan == "float64" else np.float32   # the type of r, g, b (and of everything computed from them)
    f0, f255 = T(0.0), T(255.0)
    for y in range(h):
        rev = serpentine and (y % 2 == 1)
        xs = range(w - 1, -1, -1) if rev else range(w)
        xdir = -1 if rev else 1
        for x in xs:
            r, g, b = (min(max(T(work[y, x, c]), f0), f255) for c in range(3))
            best, best_dist = 0, 1e20
            for i in range(pal.shape[0]):
                dr, dg, db = r - T(pal[i, 0]), g - T(pal[i, 1]), b - T(pal[i, 2])
                dist = dr * dr + dg * dg + db * db  # numpy scalars of type T: each product and sum rounded to T
                if float(dist) < best_dist:
                    best_dist, best = float(dist), i
            pick[y, x] = best
            c0, c1, c2 = pal[best]
            work[y, x] = (c0, c1, c2)
            err = (r - T(c0), g - T(c1), b - T(c2))
            for k, (tdx, tdy, _) in enumerate(taps):
                nx, ny = x + tdx * xdir, y + tdy
                if 0 <= nx < w and 0 <= ny < h:
                    wgt = np.float64(weights[k]) / np.float64(divisor)
                    for c in range(3):
                        work[ny, nx, c] = np.float32(np.float64(work[ny, nx, c]) + np.float64(err[c]) * wgt)
    return np.asarray(out_colors, np.uint8)[pick]


def hybrid_numba_u8(arr, pal_f32, out_colors, lut_in, lum_factor=1.0, col_factor=0.2):
    """HybridDitherStrategy's numba branch (_hybrid_numba, dithering_lib.py:1396-1494) restated in C, typed per numba's
    unification rule (float64 r / g / b, float64 scan, float64 error and luminance split).  PARITY UNPINNED, fixtures pending."""
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    h, w, _ = arr.shape
    out = np.empty_like(arr)
    rc = lib().orc_hybrid_numba_u8(_p(arr), _p(out), h, w, _p(pal_f32), pal_f32.shape[0], _p(out_colors), _p(lut_in),
                                   float(lum_factor), float(col_factor))
    if rc != 0:
        raise ValueError("oracle hybrid_numba_u8 failed")
    return out


def hybrid_numba_numpy(arr, pal_f32, out_colors, lut_in, lum_factor=1.0, col_factor=0.2):
    """A second, independent statement of the same lines with numpy scalars doing the arithmetic (every variable np.float64,
    the work array np.float32).  Slow: small images only."""
    arr = np.asarray(arr, np.uint8)
    h, w, _ = arr.shape
    work = (lut_in[arr] if lut_in is not None else arr).astype(np.float32)
    pal = np.asarray(pal_f32, np.float32)
    lf, cf = np.float64(lum_factor), np.float64(col_factor)
    F = np.float64
    pick = np.zeros((h, w), np.int64)
    for y in range(h):
        for x in range(w):
            r, g, b = (min(max(F(work[y, x, c]), F(0.0)), F(255.0)) for c in range(3))
            best, best_dist = 0, 1e20
            for i in range(pal.shape[0]):
                dr, dg, db = r - F(pal[i, 0]), g - F(pal[i, 1]), b - F(pal[i, 2])
                dist = dr * dr + dg * dg + db * db
                if float(dist) < best_dist:
                    best_dist, best = float(dist), i
            pick[y, x] = best
            c0, c1, c2 = pal[best]
            work[y, x] = (c0, c1, c2)
            err0, err1, err2 = r - F(c0), g - F(c1), b - F(c2)
            lum_err_val = F(0.299) * err0 + F(0.587) * err1 + F(0.114) * err2
            lum = (F(0.299) * lum_err_val, F(0.587) * lum_err_val, F(0.114) * lum_err_val)
            fe = tuple(lf * lc + cf * (e - lc) for lc, e in zip(lum, (err0, err1, err2)))
            for dx, dy, wgt in ((1, 0, 7.0 / 16.0), (-1, 1, 3.0 / 16.0), (0, 1, 5.0 / 16.0), (1, 1, 1.0 / 16.0)):
                nx, ny = x + dx, y + dy
                if 0 <= nx < w and 0 <= ny < h:
                    for c in range(3):
                        work[ny, nx, c] = np.float32(F(work[ny, nx, c]) + fe[c] * F(wgt))
    return np.asarray(out_colors, np.uint8)[pick]


def uniform_filter_f32(a, size):
    """scipy.ndimage.uniform_filter(a, size, mode='nearest') for a 2-D float32 array, restated: axis 0 then
    axis 1, each pass a double running sum written back as float32 (NI_UniformFilter1D)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    h, w = a.shape
    t = np.empty_like(a)
    lib().orc_uniform_filter1d_f32(_p(a), _p(t), w, h, 1, w, int(size))      # along axis 0: one line per column
    o = np.empty_like(a)
    lib().orc_uniform_filter1d_f32(_p(t), _p(o), h, w, w, 1, int(size))      # along axis 1: one line per row
    return o


def variance_gate(src_u8, var_threshold=300.0, window_radius=1):
    """AdaptiveVarianceDitherStrategy: var_map >= threshold (dithering_lib.py:988-992, 1019-1025), uint8 map"""
    px = src_u8.astype(np.float32)
    gray = (np.float32(0.299) * px[:, :, 0] + np.float32(0.587) * px[:, :, 1]) + np.float32(0.114) * px[:, :, 2]
    size = 2 * int(window_radius) + 1
    mean_sq = uniform_filter_f32(gray * gray, size)
    sq_mean = uniform_filter_f32(gray, size)
    var = np.maximum(np.float32(0.0), mean_sq - sq_mean * sq_mean)
    return np.ascontiguousarray((var >= np.float32(var_threshold)).astype(np.uint8)), var


_OSTRO = None


def ostromoukhov_coefficients():
    """float32 c_k/(c0+c1+c2) per intensity (table data: tests/golden/small.npz 'ostro_table',
    dithering_lib.py:1170-1203)."""
    global _OSTRO
    if _OSTRO is None:
        t = np.load(os.path.join(_HERE, "..", "tests", "golden", "small.npz"))["ostro_table"].astype(np.int64)
        _OSTRO = np.ascontiguousarray((t / t.sum(1, keepdims=True)).astype(np.float32))
    return _OSTRO


def var_diffusion_u8(arr, pal_f32, out_colors, lut_in, model, p0=0.0, p1=0.0, serpentine=False, gate=None):
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    h, w, _ = arr.shape
    out = np.empty_like(arr)
    coef = ostromoukhov_coefficients() if model == 4 else None
    rc = lib().orc_var_diffusion_u8(_p(arr), _p(out), h, w, _p(pal_f32), pal_f32.shape[0], _p(out_colors), _p(lut_in),
                                    model, float(p0), float(p1), 1 if serpentine else 0, _p(gate), _p(coef))
    if rc != 0:
        raise ValueError("oracle var_diffusion_u8 failed")
    return out


def apply_dithering(arr, palette, mode="bayer", params=None, use_gamma=False, y0=0, x0=0):
    """uint8 HWC -> uint8 HWC; mirrors ImageDitherer.apply_dithering (dithering_lib.py:1952-1992)
    for the in-scope modes.  (y0,x0) are the global coordinates of arr[0,0] for tile shards."""
    p = dict(MODE_DEFAULTS[mode])
    p.update(params or {})
    pal_f32, out_colors, lut_in = prepare_palette(palette, use_gamma)
    if mode == "none":
        return ordered_u8(arr, pal_f32, out_colors, lut_in, "none", y0=y0, x0=x0)
    if mode == "bayer":
        return ordered_u8(arr, pal_f32, out_colors, lut_in, "matrix", thr=bayer_matrix(p["size"]), y0=y0, x0=x0)
    if mode == "polka_dot":
        return ordered_u8(arr, pal_f32, out_colors, lut_in, "matrix", thr=polka_dot_matrix(p["tile_size"], p["gamma"]),
                          y0=y0, x0=x0)
    if mode == "blue_noise":
        return ordered_u8(arr, pal_f32, out_colors, lut_in, "matrix", thr=blue_noise(p["size"], p["seed"]),
                          y0=y0, x0=x0)
    if mode == "IGN":
        return ordered_u8(arr, pal_f32, out_colors, lut_in, "ign", scale=float(p["scale"]), seed=int(p["seed"]),
                          y0=y0, x0=x0)
    if mode == "error_diffusion":
        return error_diffusion_u8(arr, pal_f32, out_colors, lut_in, p["variant"], p["serpentine"] == "true")
    if mode == "perceptual":
        return var_diffusion_u8(arr, pal_f32, out_colors, lut_in, 1)
    if mode == "hybrid":
        return var_diffusion_u8(arr, pal_f32, out_colors, lut_in, 2, p["lum_factor"], p["col_factor"])
    if mode == "adaptive_variance":
        src = lut_in[arr] if lut_in is not None else arr
        gate, _ = variance_gate(src, p["var_threshold"], p["window_radius"])
        return var_diffusion_u8(arr, pal_f32, out_colors, lut_in, 3, gate=gate)
    if mode == "ostromoukhov":
        return var_diffusion_u8(arr, pal_f32, out_colors, lut_in, 4, serpentine=(p["serpentine"] == "true"))
    raise ValueError(f"oracle: mode {mode!r} not in scope")


# ----------------------------------------------------------------------------- palettes
def generate_uniform_palette(n):
    """dithering_lib.py:1859-1872."""
    c = []
    cube = int(math.ceil(n ** (1 / 3)))
    for r in range(cube):
        for g in range(cube):
            for b in range(cube):
                if len(c) >= n:
                    break
                rr = int(r * 255 / (cube - 1)) if cube > 1 else 128
                gg = int(g * 255 / (cube - 1)) if cube > 1 else 128
                bb = int(b * 255 / (cube - 1)) if cube > 1 else 128
                c.append((rr, gg, bb))
    return c[:n]


# ----------------------------------------------------------------------------- k-means
def kmeans_step(px, centers, mean=None):
    """one Lloyd pass: (int64 sums [K,3], int64 counts [K], inertia).  mean (float64 [3], the data mean KMeans.fit
    subtracts): equidistant samples are labelled by sklearn's own float64 expression (orc_kmeans_step_sk); None:
    lowest index by direct distances."""
    px = np.ascontiguousarray(px, dtype=np.uint8).reshape(-1, 3)
    centers = np.ascontiguousarray(centers, dtype=np.float64)
    K = centers.shape[0]
    sums = np.zeros((K, 3), np.int64)
    counts = np.zeros(K, np.int64)
    inertia = C.c_double(0)
    if mean is not None:
        mean = np.ascontiguousarray(mean, dtype=np.float64).reshape(3)
    lib().orc_kmeans_step_sk(_p(px), px.shape[0], _p(centers), K, _p(mean) if mean is not None else None, _p(sums),
                             _p(counts), C.byref(inertia))
    return sums, counts, inertia.value


def data_mean(px):
    """X.mean(axis=0) of the uint8 samples as KMeans.fit computes it: the column sums are integers below 2^53, so every
    summation order gives sum / n rounded once."""
    px = np.asarray(px).reshape(-1, 3)
    return px.sum(axis=0, dtype=np.int64).astype(np.float64) / float(px.shape[0])


def kmeans_plusplus(px, K, rs, return_indices=False):
    """sklearn.cluster._kmeans._kmeans_plusplus restated on the uint8 pixel sample (f64 arithmetic,
    direct (x-c)^2 distances); rs is a numpy RandomState.  Pinned by the km*_init_idx fixtures (sklearn's own picks
    for RandomState(42) on the three reference k-means inputs, tests/golden/make_golden.py)."""
    X = np.asarray(px, dtype=np.float64).reshape(-1, 3)
    n = X.shape[0]
    trials = 2 + int(np.log(K))
    centers = np.empty((K, 3), np.float64)
    w = np.ones(n, np.float64)
    cid = rs.choice(n, p=w / w.sum())  # sklearn: choice(n_samples, p=sample_weight / sample_weight.sum())
    ids = [int(cid)]
    centers[0] = X[cid]
    closest = ((X - centers[0]) ** 2).sum(1)
    pot = closest.sum()
    for c in range(1, K):
        rv = rs.uniform(size=trials) * pot
        cand = np.searchsorted(np.cumsum(closest), rv)
        np.clip(cand, None, n - 1, out=cand)
        dc = ((X[cand][:, None, :] - X[None, :, :]) ** 2).sum(2)
        np.minimum(closest, dc, out=dc)
        pots = dc.sum(1)
        b = int(np.argmin(pots))
        pot = pots[b]
        closest = dc[b]
        centers[c] = X[cand[b]]
        ids.append(int(cand[b]))
    return (centers, np.array(ids)) if return_indices else centers


def kmeans_lloyd(px, init_centers, max_iter=300, tol=1e-4, sklearn_ties=True):
    """sklearn _kmeans_single_lloyd semantics on exact integer sums:
    stop on unchanged labels (detected through unchanged sums/counts) or squared centre shift
    <= tol * mean(var(X)); empty clusters keep their previous centre.  sklearn_ties: equidistant samples are labelled
    by sklearn's float64 expression on the mean-centred data (orc_kmeans_step_sk) - with it the 11 sklearn fixtures
    (km*, kmx_*) are reproduced to 2e-13 with equal iteration counts; without it (lowest index) three of them drift."""
    px = np.ascontiguousarray(px, dtype=np.uint8).reshape(-1, 3)
    X = px.astype(np.float64)
    tol_abs = float(np.mean(np.var(X, axis=0)) * tol)
    centers = np.array(init_centers, dtype=np.float64)
    mean = data_mean(px) if sklearn_ties else None
    prev = None
    n_iter = 0
    for n_iter in range(1, max_iter + 1):
        sums, counts, _ = kmeans_step(px, centers, mean)
        new = centers.copy()
        nz = counts > 0
        new[nz] = sums[nz] / counts[nz, None]
        shift = float(((new - centers) ** 2).sum())
        same = prev is not None and np.array_equal(prev[0], sums) and np.array_equal(prev[1], counts)
        centers = new
        prev = (sums, counts)
        if same or shift <= tol_abs:
            break
    _, _, inertia = kmeans_step(px, centers)
    return centers, inertia, n_iter


# ----------------------------------------------------------------------------- inputs
def few_colour_pixels(n, nc, seed):
    """n pixels drawn from nc random colours (tests/golden/make_golden.py: KM_FEW) -> uint8 [n,3]"""
    rs = np.random.RandomState(seed)
    cols = rs.randint(0, 256, (nc, 3))
    return cols[rs.randint(0, nc, n)].astype(np.uint8)


def rnd(h, w, seed):
    return np.random.RandomState(seed).randint(0, 256, (h, w, 3), dtype=np.uint8)


def grad(h, w):
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([x % 256, y % 256, ((x + y) // 2) % 256], -1).astype(np.uint8)


def imgl(h, w, seed, kind="smooth"):
    """Image-like content without transcendental functions (identical on every machine): ramps, a little structure and
    grain; "dark" keeps every channel within ~45 levels.  A palette extracted from it crowds a small part of the cube."""
    y, x = np.mgrid[0:h, 0:w]
    rs = np.random.RandomState(seed)
    if kind == "dark":
        chans = [20 + (35 * x) // w + (10 * y) // h, 18 + (25 * y) // h + (x // 9) % 3, 25 + (30 * ((x + 2 * y) % w)) // w]
        grain = rs.randint(-2, 3, (h, w, 3))
    else:
        chans = [60 + (100 * x) // w + (30 * y) // h, 90 + (80 * y) // h + (x // 7) % 9, 150 + (60 * (x + y)) // (w + h) - (y // 5) % 7]
        grain = rs.randint(-3, 4, (h, w, 3))
    return np.clip(np.stack(chans, -1) + grain, 0, 255).astype(np.uint8)


def palr(K, seed=7):
    return [tuple(int(v) for v in c) for c in np.random.RandomState(seed).randint(0, 256, (K, 3))]


def H(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


# ----------------------------------------------------------------------------- scipy statement
def ordered_scipy(arr, palette, thr, use_gamma=False, workers=-1):
    """The matrix strategy stated directly with the third-party calls the reference makes
    (dithering_lib.py:355-378 inside the uint8 wrapper :1952-1992): scipy.spatial.KDTree.query(k=2,
    workers=-1) + numpy.  Independent of dp_oracle.c; used as a cross-check of the C restatement and as the
    multi-core CPU baseline of BASELINE.md section 3 (item 2)."""
    from scipy.spatial import KDTree
    arr = np.asarray(arr, dtype=np.uint8)
    h, w, _ = arr.shape
    pal_f32, out_colors, lut_in = prepare_palette(palette, use_gamma)
    src = lut_in[arr] if lut_in is not None else arr
    px = src.reshape(-1, 3).astype(np.float32)
    d, idx = KDTree(pal_f32).query(px, k=2, workers=workers)
    s = d ** 2
    tot = s[:, 0] + s[:, 1]
    with np.errstate(invalid="ignore", divide="ignore"):
        f = np.where(tot == 0, 0.0, s[:, 0] / tot)
    th_h, th_w = thr.shape
    tiled = np.tile(thr, ((h + th_h - 1) // th_h, (w + th_w - 1) // th_w))[:h, :w].reshape(-1)
    pick = np.where(f <= tiled, idx[:, 0], idx[:, 1])
    return out_colors[pick].reshape(h, w, 3)
