"""Drop-in counterpart of dither_pie's `dithering_lib` for the hot path, on MI355X.

Same public names, constructor arguments, parameter metadata and error behaviour as the reference
module (reference lines cited per item), so `from dither_pie_amd.dithering_lib import ImageDitherer,
DitherMode, ColorReducer` can replace `from dithering_lib import ...` in dither_cli.py /
dither_pie_gui.py / video_processor.py.  Pixel work happens in libditherpie_hip.so on the GPU; this
file is host plumbing only (palette preparation, parameter handling, tensor hand-off).  There is no
CPU fallback: without the shared library or a HIP device the calls raise DitherPieError.

In scope: none, bayer, blue_noise, IGN, error_diffusion, polka_dot, perceptual,
hybrid, adaptive_variance, ostromoukhov; k-means / uniform / median-cut palettes.  The other DitherMode members exist for configuration compatibility and raise
NotImplementedError when used.

Extras that the reference does not have (all optional): ImageDitherer.apply_dithering_frames() for
batches of frames already resident in HBM, and tile offsets (y0, x0) for row-band sharding.
"""
from __future__ import annotations

import os

import math
from collections import OrderedDict
from enum import Enum
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from . import _tables

_staging = None  # threading.local(): .bufs = {nbytes: (pinned in, pinned out)}


def _pinned_pair(nbytes: int):
    """Two page-locked uint8 host buffers of `nbytes` for the calling thread (kept: at most 4 sizes per thread)."""
    import threading
    import torch
    global _staging
    if _staging is None:
        _staging = threading.local()
    bufs = getattr(_staging, "bufs", None)
    if bufs is None:
        bufs = _staging.bufs = OrderedDict()
    pair = bufs.get(nbytes)
    if pair is None:
        pair = (torch.empty(nbytes, dtype=torch.uint8, pin_memory=True),
                torch.empty(nbytes, dtype=torch.uint8, pin_memory=True))
        bufs[nbytes] = pair
        while len(bufs) > 4:
            bufs.popitem(last=False)
    else:
        bufs.move_to_end(nbytes)
    return pair

def _pil_rgbx_into(rgb, host_u8) -> bool:
    """The pixels of a PIL 'RGB' image in Pillow's own four-bytes-per-pixel layout, written into `host_u8` (a uint8 numpy view of
    h * w * 4 bytes) in pieces, with no intermediate bytes object of the whole image: what Image.tobytes("raw", "RGBX") does,
    minus its b"".join and the caller's copy.  False when the (private, long-stable) encoder interface is not there."""
    try:
        from PIL import Image
        rgb.load()
        enc = Image._getencoder(rgb.mode, "raw", "RGBX")
        enc.setimage(rgb.im, (0, 0) + rgb.size)
        pos, total = 0, host_u8.shape[0]
        piece = max(1 << 20, rgb.size[0] * 4)   # (the raw encoder emits whole rows: RawEncode.c)
        while True:
            _, errcode, data = enc.encode(piece)
            n = len(data)
            if pos + n > total:
                return False
            host_u8[pos:pos + n] = np.frombuffer(data, dtype=np.uint8)
            pos += n
            if errcode:
                break
        return errcode > 0 and pos == total
    except Exception:  # noqa: BLE001 - any surprise: the packed path
        return False


__all__ = [
    "DitherMode", "PixelizeMethod", "PaletteSource", "ImageDitherer", "ColorReducer", "DitherUtils",
    "BaseDitherStrategy", "ErrorDiffusionKernel", "NoDitherStrategy", "MatrixDitherStrategy",
    "BayerDitherStrategy", "BlueNoiseDitherStrategy", "InterleavedGradientNoiseDitherStrategy",
    "ErrorDiffusionDitherStrategy", "PolkaDotDitherStrategy", "PerceptualDitherStrategy", "HybridDitherStrategy",
    "AdaptiveVarianceDitherStrategy", "OstromoukhovDitherStrategy", "generate_blue_noise",
]


# ------------------------------------------------------------------------------------- enums
class DitherMode(Enum):
    """dithering_lib.py:61-75 (same member names and string values, incl. the upper-case "IGN")."""
    NONE = "none"
    BAYER = "bayer"
    ERROR_DIFFUSION = "error_diffusion"
    RIEMERSMA = "riemersma"
    BLUE_NOISE = "blue_noise"
    INTERLEAVED_GRADIENT_NOISE = "IGN"
    POLKA_DOT = "polka_dot"
    WAVELET = "wavelet"
    ADAPTIVE_VARIANCE = "adaptive_variance"
    PERCEPTUAL = "perceptual"
    HYBRID = "hybrid"
    HALFTONE = "halftone"
    OSTROMOUKHOV = "ostromoukhov"


class PixelizeMethod(Enum):
    """dithering_lib.py:78-82"""
    NONE = "none"
    REGULAR = "regular"
    NEURAL = "neural"


class PaletteSource(Enum):
    """dithering_lib.py:85-91"""
    MEDIAN_CUT = "median_cut"
    KMEANS = "kmeans"
    UNIFORM = "uniform"
    CUSTOM = "custom"
    FROM_FILE = "file"


_OUT_OF_SCOPE = {DitherMode.RIEMERSMA, DitherMode.WAVELET, DitherMode.HALFTONE}
# per-pixel independent given global coordinates: these shard by row bands / tiles (sharding.dither_band); the diffusers do not
ORDERED_MODES = {DitherMode.NONE, DitherMode.BAYER, DitherMode.BLUE_NOISE, DitherMode.INTERLEAVED_GRADIENT_NOISE, DitherMode.POLKA_DOT}


# ------------------------------------------------------------------------------------- tap tables
def _kernel(taps, divisor, description, rows):
    return {"weights": taps, "divisor": divisor, "description": description, "rows": rows}


class ErrorDiffusionKernel:
    """Error-diffusion tap tables: (dx, dy, weight) lists and divisors (dithering_lib.py:96-209)."""

    FLOYD_STEINBERG = _kernel([(1, 0, 7), (-1, 1, 3), (0, 1, 5), (1, 1, 1)], 16,
                              "Classic Floyd-Steinberg (4 neighbors)", 2)
    JJN = _kernel([(1, 0, 7), (2, 0, 5)]
                  + [(dx, 1, wt) for dx, wt in zip(range(-2, 3), (3, 5, 7, 5, 3))]
                  + [(dx, 2, wt) for dx, wt in zip(range(-2, 3), (1, 3, 5, 3, 1))], 48,
                  "Jarvis-Judice-Ninke (12 neighbors, smooth gradients)", 3)
    STUCKI = _kernel([(1, 0, 8), (2, 0, 4)]
                     + [(dx, 1, wt) for dx, wt in zip(range(-2, 3), (2, 4, 8, 4, 2))]
                     + [(dx, 2, wt) for dx, wt in zip(range(-2, 3), (1, 2, 4, 2, 1))], 42,
                     "Stucki (12 neighbors, photographic quality)", 3)
    BURKES = _kernel([(1, 0, 8), (2, 0, 4)]
                     + [(dx, 1, wt) for dx, wt in zip(range(-2, 3), (2, 4, 8, 4, 2))], 32,
                     "Burkes (7 neighbors, fast)", 2)
    ATKINSON = _kernel([(1, 0, 1), (2, 0, 1), (-1, 1, 1), (0, 1, 1), (1, 1, 1), (0, 2, 1)], 8,
                       "Atkinson (6 neighbors, classic Mac look)", 3)
    SIERRA = _kernel([(1, 0, 5), (2, 0, 3)]
                     + [(dx, 1, wt) for dx, wt in zip(range(-2, 3), (2, 4, 5, 4, 2))]
                     + [(dx, 2, wt) for dx, wt in zip(range(-1, 2), (2, 3, 2))], 32,
                     "Sierra Full (10 neighbors, high quality)", 3)
    SIERRA_TWO_ROW = _kernel([(1, 0, 4), (2, 0, 3)]
                             + [(dx, 1, wt) for dx, wt in zip(range(-2, 3), (1, 2, 3, 2, 1))], 16,
                             "Sierra Two-Row (8 neighbors, balanced)", 2)
    SIERRA_LITE = _kernel([(1, 0, 2), (-1, 1, 1), (0, 1, 1)], 4, "Sierra Lite (4 neighbors, fastest)", 2)

    _NAMES = ("floyd_steinberg", "jjn", "stucki", "burkes", "atkinson", "sierra", "sierra_two_row",
              "sierra_lite")

    @classmethod
    def get_kernel(cls, name: str) -> Dict[str, Any]:
        """Unknown names fall back to Floyd-Steinberg (dithering_lib.py:203)."""
        if name in cls._NAMES:
            return getattr(cls, name.upper())
        return cls.FLOYD_STEINBERG

    @classmethod
    def list_kernels(cls) -> List[str]:
        return list(cls._NAMES)


# ------------------------------------------------------------------------------------- threshold tables
def _table(den, rows):
    return (np.array(rows, dtype=np.float64) / den).astype(np.float32)


class DitherUtils:
    """Threshold tables (stored as integer numerators over a power of two; the float32 values equal
    dithering_lib.py:1705-1768 bit for bit, including the two non-canonical entries at the end of
    BAYER8x8 row 3) and the gamma helpers (dithering_lib.py:1788-1802)."""

    BAYER2x2 = _table(4, [[1, 3], [4, 2]])
    BAYER4x4 = _table(32, [[1, 17, 5, 21], [25, 9, 29, 13], [7, 23, 3, 19], [31, 15, 27, 11]])
    BAYER8x8 = _table(64, [
        [1, 33, 9, 41, 3, 35, 11, 43], [49, 17, 57, 25, 51, 19, 59, 27],
        [13, 45, 5, 37, 15, 47, 7, 39], [61, 29, 53, 21, 63, 31, 54, 22],
        [4, 36, 12, 44, 2, 34, 10, 42], [52, 20, 60, 28, 50, 18, 58, 26],
        [16, 48, 8, 40, 14, 46, 6, 38], [64, 32, 56, 24, 62, 30, 54, 22]])
    BAYER16x16 = _table(256, [
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
        [256, 128, 224, 96, 248, 120, 216, 88, 254, 126, 222, 94, 246, 118, 214, 86]])
    PSX4x4 = _table(16, [[1, 9, 3, 11], [13, 5, 15, 7], [3, 11, 1, 9], [15, 7, 13, 5]])

    _BY_SIZE = {"2x2": "BAYER2x2", "4x4": "BAYER4x4", "8x8": "BAYER8x8", "16x16": "BAYER16x16",
                "psx4x4": "PSX4x4", "psx": "PSX4x4"}

    @staticmethod
    def get_threshold_matrix(mode: DitherMode, size: str = "4x4") -> np.ndarray:
        """dithering_lib.py:1770-1786"""
        if mode == DitherMode.NONE:
            return np.ones((1, 1), dtype=np.float32)
        if mode == DitherMode.BAYER:
            return getattr(DitherUtils, DitherUtils._BY_SIZE.get(size, "BAYER4x4"))
        raise ValueError(f"Unsupported matrix mode: {mode}")

    @staticmethod
    def srgb_to_linear(c: np.ndarray) -> np.ndarray:
        """dithering_lib.py:1788-1794"""
        c = np.asarray(c)
        out = np.empty_like(c, dtype=np.float32)
        lo = c <= 0.04045
        out[lo] = c[lo] / 12.92
        out[~lo] = ((c[~lo] + 0.055) / 1.055) ** 2.4
        return out

    @staticmethod
    def linear_to_srgb(c: np.ndarray) -> np.ndarray:
        """dithering_lib.py:1796-1802"""
        c = np.asarray(c)
        out = np.empty_like(c, dtype=np.float32)
        lo = c <= 0.0031308
        out[lo] = c[lo] * 12.92
        out[~lo] = 1.055 * (c[~lo] ** (1.0 / 2.4)) - 0.055
        return out


# ------------------------------------------------------------------------------------- device-object caches
class _LRU(OrderedDict):
    """Small LRU of device objects; get_or_make is atomic (the GUI calls the ditherer from several threads)."""

    def __init__(self, cap):
        super().__init__()
        self.cap = cap
        import threading
        self._lock = threading.RLock()

    def get_or_make(self, key, make):
        with self._lock:
            if key in self:
                self.move_to_end(key)
                return self[key]
            val = make()
            self[key] = val
            while len(self) > self.cap:
                self.popitem(last=False)
            return val


# device handles are process-local and never stored on the (picklable) ditherer objects
_PALETTES = _LRU(32)
_THRESHOLDS = _LRU(32)


def drop_device_caches():
    """Forget every cached device palette / threshold matrix (they are re-created on demand)."""
    with _PALETTES._lock:
        _PALETTES.clear()
    with _THRESHOLDS._lock:
        _THRESHOLDS.clear()
    OstromoukhovDitherStrategy._coef_cache.clear()


def _device_index():
    import torch
    return torch.cuda.current_device() if torch.cuda.is_available() else -1


def _device_palette(pal_f32, out_colors, lut_in):
    from . import backend
    key = (_device_index(), pal_f32.tobytes(), out_colors.tobytes(), None if lut_in is None else lut_in.tobytes())
    return _PALETTES.get_or_make(key, lambda: backend.Palette(pal_f32, out_colors, lut_in))


def _device_thresholds_matrix(matrix):
    from . import backend
    m = np.ascontiguousarray(matrix, dtype=np.float32)
    key = (_device_index(), "m", m.shape, m.tobytes())
    return _THRESHOLDS.get_or_make(key, lambda: backend.Thresholds.from_matrix(m))


def _device_thresholds_blue(size, seed):
    from . import backend
    key = (_device_index(), "bn", int(size), int(seed))
    return _THRESHOLDS.get_or_make(key, lambda: backend.Thresholds.blue_noise(size, seed))


def prepare_palette(palette, use_gamma):
    """Host-side palette conversion of ImageDitherer.apply_dithering.

    -> (pal_f32 [K,3]: what the nearest-colour search sees (dithering_lib.py:1970-1974),
        out_colors [K,3] uint8: the bytes a chosen entry becomes (:1984-1990),
        lut_in: uint8[256] applied to the image first (:1957-1959), or None)"""
    pal = np.array(palette, dtype=np.float32).reshape(-1, 3)
    if not use_gamma:
        return np.ascontiguousarray(pal), np.ascontiguousarray(pal.astype(np.uint8)), None
    ints = pal.astype(np.int64)
    if np.array_equal(ints.astype(np.float32), pal) and ints.min() >= 0 and ints.max() <= 255:
        pal_lin = _tables.PAL_LIN[ints]
    else:  # values outside the uint8 grid: evaluate the reference's formula directly
        pal_lin = np.clip(DitherUtils.srgb_to_linear(pal / 255.0) * 255.0, 0, 255).astype(np.float32)
    out_colors = _tables.LUT_OUT[pal_lin.astype(np.uint8)]
    return np.ascontiguousarray(pal_lin, dtype=np.float32), np.ascontiguousarray(out_colors), _tables.LUT_IN


def _index_palette(palette_arr):
    """A device palette whose output bytes encode the chosen index (for the strategy-level API)."""
    pal = np.ascontiguousarray(palette_arr, dtype=np.float32).reshape(-1, 3)
    K = pal.shape[0]
    idx = np.arange(K)
    enc = np.stack([idx & 255, idx >> 8, np.zeros_like(idx)], axis=1).astype(np.uint8)
    return _device_palette(pal, enc, None)


def _pixels_to_frame(pixels, image_size):
    import torch
    h, w = image_size
    px = np.asarray(pixels)
    if px.shape != (h * w, 3):
        raise ValueError(f"pixels must have shape ({h * w}, 3)")
    u8 = px.astype(np.uint8)
    if not np.array_equal(u8.astype(px.dtype), px):
        raise ValueError("the MI355X backend dithers uint8 images: pixel values must be integers in [0, 255]")
    return torch.from_numpy(np.ascontiguousarray(u8.reshape(h, w, 3))).cuda()


def _decode(out_frame, palette_arr):
    o = out_frame.cpu().numpy().reshape(-1, 3)
    idx = o[:, 0].astype(np.int64) | (o[:, 1].astype(np.int64) << 8)
    return np.asarray(palette_arr)[idx, :]


# ------------------------------------------------------------------------------------- strategies
class BaseDitherStrategy:
    """dithering_lib.py:313-330: dither(pixels [N,3] f32, palette_arr [K,3] f32, (h, w)) -> [N,3]."""

    def dither(self, pixels: np.ndarray, palette_arr: np.ndarray, image_size: Tuple[int, int]) -> np.ndarray:
        raise NotImplementedError

    @staticmethod
    def get_parameter_info() -> Optional[Dict[str, Any]]:
        return None

    def get_current_parameters(self) -> Dict[str, Any]:
        return {}

    # device-level entry used by ImageDitherer: uint8 frames in HBM -> uint8 frames in HBM
    def _run(self, frames, pal, y0=0, x0=0, out=None):
        raise NotImplementedError


class NoDitherStrategy(BaseDitherStrategy):
    """Nearest palette colour (dithering_lib.py:333-341)."""

    def _run(self, frames, pal, y0=0, x0=0, out=None):
        from . import backend
        return backend.ordered(frames, pal, backend.MODE_NEAREST, y0=y0, x0=x0, out=out)

    def dither(self, pixels, palette_arr, image_size):
        out = self._run(_pixels_to_frame(pixels, image_size), _index_palette(palette_arr))
        return _decode(out, palette_arr)


class MatrixDitherStrategy(BaseDitherStrategy):
    """Threshold-matrix dithering between the two nearest colours (dithering_lib.py:346-378)."""

    def __init__(self, threshold_matrix: np.ndarray):
        self.threshold_matrix = threshold_matrix

    def _thresholds(self):
        return _device_thresholds_matrix(self.threshold_matrix)

    def _run(self, frames, pal, y0=0, x0=0, out=None):
        from . import backend
        return backend.ordered(frames, pal, backend.MODE_MATRIX, thr=self._thresholds(), y0=y0, x0=x0, out=out)

    def dither(self, pixels, palette_arr, image_size):
        out = self._run(_pixels_to_frame(pixels, image_size), _index_palette(palette_arr))
        return _decode(out, palette_arr)


def generate_blue_noise(size: int = 64, seed: int = 42) -> np.ndarray:
    """Void-filling blue-noise matrix (dithering_lib.py:381-399), generated on the GPU."""
    return _device_thresholds_blue(size, seed).numpy()


class BayerDitherStrategy(MatrixDitherStrategy):
    """dithering_lib.py:402-448"""

    @staticmethod
    def get_parameter_info() -> Dict[str, Any]:
        return {
            "size": {
                "type": "choice",
                "default": "4x4",
                "choices": ["2x2", "4x4", "8x8", "16x16", "psx4x4"],
                "label": "Matrix",
                "description": "Bayer matrix size or PSX 4x4 variant (larger = finer patterns)",
            }
        }

    def __init__(self, size: str = "4x4"):
        self.size = size
        super().__init__(DitherUtils.get_threshold_matrix(DitherMode.BAYER, size))

    def get_current_parameters(self) -> Dict[str, Any]:
        return {"size": self.size}


class BlueNoiseDitherStrategy(MatrixDitherStrategy):
    """dithering_lib.py:451-499 (the matrix is generated and cached on the device per (size, seed))."""

    @staticmethod
    def get_parameter_info() -> Dict[str, Any]:
        return {
            "size": {
                "type": "int", "default": 64, "min": 32, "max": 128, "label": "Matrix Size",
                "description": "Size of the blue noise matrix (larger = more detail but slower)",
            },
            "seed": {
                "type": "int", "default": 42, "min": 0, "max": 9999, "label": "Random Seed",
                "description": "Seed for noise generation (different seeds = different patterns)",
            },
        }

    def __init__(self, size: int = 64, seed: int = 42):
        self.size = size
        self.seed = seed
        self._matrix = None

    @property
    def threshold_matrix(self):
        if self._matrix is None:
            self._matrix = generate_blue_noise(self.size, self.seed)
        return self._matrix

    def _thresholds(self):
        return _device_thresholds_blue(self.size, self.seed)

    def get_current_parameters(self) -> Dict[str, Any]:
        return {"size": self.size, "seed": self.seed}

    def __getstate__(self):
        return {"size": self.size, "seed": self.seed, "_matrix": None}


class PolkaDotDitherStrategy(MatrixDitherStrategy):
    """Circular-dot threshold tile (dithering_lib.py:695-766): the decision rule is MatrixDitherStrategy's,
    only the tile differs; the tile_size x tile_size matrix is host setup exactly as in the reference."""

    @staticmethod
    def get_parameter_info() -> Dict[str, Any]:
        return {
            "tile_size": {
                "type": "int", "default": 8, "min": 4, "max": 32, "label": "Tile Size",
                "description": "Size of the repeating dot pattern",
            },
            "gamma": {
                "type": "float", "default": 1.5, "min": 0.5, "max": 3.0, "step": 0.1, "label": "Gamma",
                "description": "Controls dot shape curve (higher = sharper edges)",
            },
        }

    def __init__(self, tile_size: int = 8, gamma: float = 1.5):
        self.tile_size = tile_size
        self.gamma = gamma
        super().__init__(self._generate_polka_dot_matrix(tile_size, gamma))

    def get_current_parameters(self) -> Dict[str, Any]:
        return {"tile_size": self.tile_size, "gamma": self.gamma}

    @staticmethod
    def _generate_polka_dot_matrix(tile_size: int, gamma: float) -> np.ndarray:
        """1 - (distance to the tile centre / corner distance)^gamma, float64 then float32
        (dithering_lib.py:733-743)."""
        c = (tile_size - 1) / 2
        xv, yv = np.meshgrid(np.arange(tile_size), np.arange(tile_size))
        norm = np.sqrt((xv - c) ** 2 + (yv - c) ** 2) / (np.sqrt(c ** 2 + c ** 2) + 1e-9)
        return np.clip(1.0 - norm ** gamma, 0, 1).astype(np.float32)


class InterleavedGradientNoiseDitherStrategy(BaseDitherStrategy):
    """dithering_lib.py:502-571"""

    @staticmethod
    def get_parameter_info() -> Dict[str, Any]:
        return {
            "scale": {
                "type": "float", "default": 1.0, "min": 0.1, "max": 10.0, "step": 0.1, "label": "Scale",
                "description": "Noise frequency (lower = larger pattern, higher = finer grain)",
            },
            "seed": {
                "type": "int", "default": 0, "min": 0, "max": 9999, "label": "Seed",
                "description": "Deterministic offset to shift the pattern",
            },
        }

    def __init__(self, scale: float = 1.0, seed: int = 0):
        self.scale = float(scale)
        self.seed = int(seed)

    def _generate_thresholds(self, image_size: Tuple[int, int]) -> np.ndarray:
        from . import backend
        h, w = image_size
        return backend.ign_thresholds(h, w, self.scale, self.seed).cpu().numpy()

    def _run(self, frames, pal, y0=0, x0=0, out=None):
        from . import backend
        return backend.ordered(frames, pal, backend.MODE_IGN, ign_scale=self.scale, ign_seed=self.seed,
                               y0=y0, x0=x0, out=out)

    def dither(self, pixels, palette_arr, image_size):
        out = self._run(_pixels_to_frame(pixels, image_size), _index_palette(palette_arr))
        return _decode(out, palette_arr)

    def get_current_parameters(self) -> Dict[str, Any]:
        return {"scale": self.scale, "seed": self.seed}


# Which of the reference's two error-diffusion arithmetics to reproduce (dithering_lib.py:638-653 picks by whether numba
# imports): "python" -- the pure-Python loop (:655-690), what the reference runs in an environment without numba, pinned by
# the golden fixtures -- or "numba" -- _error_diffusion_numba (:213-308), typed per numba's unification rule: r / g / b are
# assigned a float32 element and float64 literals (:239-251), hence float64 -- a float64 linear scan with the first minimum
# winning, a float64 error, float64 products and sums with one rounding on the store.  Restated in the oracle; fixtures
# pending (no numba in the build image), so this branch is parity-unpinned.
ERROR_DIFFUSION_ARITHMETIC = os.environ.get("DITHER_PIE_ED_ARITHMETIC", "python")


class ErrorDiffusionDitherStrategy(BaseDitherStrategy):
    """dithering_lib.py:576-690 (by default the semantics of the pure-Python branch, which is what runs when numba is
    not installed -- see ERROR_DIFFUSION_ARITHMETIC; parameters are the reference's strings)."""

    @staticmethod
    def get_parameter_info() -> Dict[str, Any]:
        return {
            "variant": {
                "type": "choice", "default": "atkinson", "choices": ErrorDiffusionKernel.list_kernels(),
                "label": "Algorithm", "description": "Error diffusion algorithm variant",
            },
            "serpentine": {
                "type": "choice", "default": "false", "choices": ["true", "false"], "label": "Serpentine Scan",
                "description": "Alternates direction each row to reduce artifacts",
            },
        }

    def __init__(self, variant: str = "atkinson", serpentine: str = "false"):
        self.variant = variant
        self.serpentine = (serpentine == "true")
        self._kernel = ErrorDiffusionKernel.get_kernel(variant)

    def get_current_parameters(self) -> Dict[str, Any]:
        return {"variant": self.variant, "serpentine": "true" if self.serpentine else "false"}

    def _run(self, frames, pal, y0=0, x0=0, out=None):
        from . import backend
        if y0 or x0:
            raise ValueError("error diffusion carries state across the whole raster and cannot be tiled")
        return backend.error_diffusion(frames, pal, self._kernel["weights"], self._kernel["divisor"],
                                       self.serpentine, out=out, arithmetic=ERROR_DIFFUSION_ARITHMETIC)

    def dither(self, pixels, palette_arr, image_size):
        out = self._run(_pixels_to_frame(pixels, image_size), _index_palette(palette_arr))
        return _decode(out, palette_arr)


class _VariableDiffuser(BaseDitherStrategy):
    """Floyd-Steinberg-shaped scans whose coefficients depend on the source pixel."""

    def _run(self, frames, pal, y0=0, x0=0, out=None):
        if y0 or x0:
            raise ValueError("error diffusion carries state across the whole raster and cannot be tiled")
        return self._diffuse(frames, pal, out)

    def dither(self, pixels, palette_arr, image_size):
        out = self._run(_pixels_to_frame(pixels, image_size), _index_palette(palette_arr))
        return _decode(out, palette_arr)


class PerceptualDitherStrategy(_VariableDiffuser):
    """Floyd-Steinberg weights scaled by 0.5 + 0.5*luminance/255 of the diffusing pixel (dithering_lib.py:1030-1066;
    the reference's optional base_weights argument is not exposed by any caller and is not supported)."""

    def _diffuse(self, frames, pal, out):
        from . import backend
        return backend.variable_diffusion(frames, pal, backend.DIFFUSER_PERCEPTUAL, out=out)


class HybridDitherStrategy(_VariableDiffuser):
    """Luminance part of the error diffused at lum_factor, colour part at col_factor (dithering_lib.py:1071-1155).  By default
    the pure-Python branch (:1127-1152), which is what runs without numba and is pinned by reference fixtures; with
    ERROR_DIFFUSION_ARITHMETIC = "numba" the strategy's numba branch (_hybrid_numba, :1396-1494, dispatched at :1114-1125),
    which clamps, scans the palette in float64 and keeps a float64 error -- typed per numba's unification rule, unpinned."""

    @staticmethod
    def get_parameter_info() -> Dict[str, Any]:
        return {
            "lum_factor": {
                "type": "float", "default": 1.0, "min": 0.0, "max": 2.0, "step": 0.1, "label": "Luminance Factor",
                "description": "Strength of luminance error diffusion (1.0 = full, 0.0 = none)",
            },
            "col_factor": {
                "type": "float", "default": 0.2, "min": 0.0, "max": 2.0, "step": 0.1, "label": "Color Factor",
                "description": "Strength of color error diffusion (lower = less color noise)",
            },
        }

    def __init__(self, lum_factor: float = 1.0, col_factor: float = 0.2):
        self.lum_factor = lum_factor
        self.col_factor = col_factor
        self.fs_offsets = [(1, 0, 7 / 16), (-1, 1, 3 / 16), (0, 1, 5 / 16), (1, 1, 1 / 16)]

    def get_current_parameters(self) -> Dict[str, Any]:
        return {"lum_factor": self.lum_factor, "col_factor": self.col_factor}

    def _diffuse(self, frames, pal, out):
        from . import backend
        if ERROR_DIFFUSION_ARITHMETIC == "numba":
            return backend.hybrid_numba(frames, pal, self.lum_factor, self.col_factor, out=out)
        return backend.variable_diffusion(frames, pal, backend.DIFFUSER_HYBRID, self.lum_factor, self.col_factor, out=out)


class AdaptiveVarianceDitherStrategy(_VariableDiffuser):
    """Floyd-Steinberg diffusion only from pixels whose local grayscale variance reaches var_threshold
    (dithering_lib.py:946-1025); the variance map reproduces scipy.ndimage.uniform_filter on the device."""

    @staticmethod
    def get_parameter_info() -> Dict[str, Any]:
        return {
            "var_threshold": {
                "type": "float", "default": 300.0, "min": 0.0, "max": 1000.0, "step": 10.0, "label": "Variance Threshold",
                "description": "Threshold for local variance to trigger error diffusion",
            },
            "window_radius": {
                "type": "int", "default": 1, "min": 1, "max": 5, "label": "Window Radius",
                "description": "Radius of window for computing local variance",
            },
        }

    def __init__(self, var_threshold: float = 300.0, window_radius: int = 1):
        self.var_threshold = var_threshold
        self.window_radius = window_radius

    def get_current_parameters(self) -> Dict[str, Any]:
        return {"var_threshold": self.var_threshold, "window_radius": self.window_radius}

    def _diffuse(self, frames, pal, out):
        from . import backend
        gate = backend.variance_gate(frames, pal, self.var_threshold, self.window_radius)
        return backend.variable_diffusion(frames, pal, backend.DIFFUSER_ADAPTIVE_VARIANCE, gate=gate, out=out)


class OstromoukhovDitherStrategy(_VariableDiffuser):
    """Ostromoukhov's intensity-dependent three-tap diffusion (dithering_lib.py:1160-1269)."""

    COEFFS_TABLE = [tuple(int(v) for v in row) for row in _tables.OSTROMOUKHOV]
    _coef_cache: Dict[int, Any] = {}

    @staticmethod
    def get_parameter_info() -> Dict[str, Any]:
        return {
            "serpentine": {
                "type": "choice", "default": "false", "choices": ["true", "false"], "label": "Serpentine Scan",
                "description": "Alternates direction each row to reduce artifacts",
            }
        }

    def __init__(self, serpentine: str = "false"):
        self.serpentine = (serpentine == "true")

    def get_current_parameters(self) -> Dict[str, Any]:
        return {"serpentine": "true" if self.serpentine else "false"}

    def _diffuse(self, frames, pal, out):
        import torch
        from . import backend
        dev = _device_index()
        coef = OstromoukhovDitherStrategy._coef_cache.get(dev)
        if coef is None:
            t = _tables.OSTROMOUKHOV
            coef = torch.from_numpy((t / t.sum(1, keepdims=True)).astype(np.float32)).cuda()  # f32(c_k / divisor)
            OstromoukhovDitherStrategy._coef_cache[dev] = coef
        return backend.variable_diffusion(frames, pal, backend.DIFFUSER_OSTROMOUKHOV, serpentine=self.serpentine,
                                          coef=coef, out=out)


# ------------------------------------------------------------------------------------- palettes
class ColorReducer:
    """Palette producers (dithering_lib.py:1807-1872)."""

    @staticmethod
    def find_dominant_channel(colors: List[Tuple[int, int, int]]) -> int:
        spans = [max(c[ch] for c in colors) - min(c[ch] for c in colors) for ch in range(3)]
        return spans.index(max(spans))

    @staticmethod
    def median_cut(colors: List[Tuple[int, int, int]], depth: int) -> List[Tuple[int, int, int]]:
        """dithering_lib.py:1822-1833: stable sort on the widest channel, split at len//2, bucket mean
        truncated per channel; an empty bucket yields (0, 0, 0)."""
        if not colors:
            return [(0, 0, 0)]
        if depth == 0:
            n = len(colors)
            return [tuple(int(sum(col) / n) for col in zip(*colors))]
        ch = ColorReducer.find_dominant_channel(colors)
        colors.sort(key=lambda c: c[ch])
        half = len(colors) // 2
        return ColorReducer.median_cut(colors[:half], depth - 1) + ColorReducer.median_cut(colors[half:], depth - 1)

    @staticmethod
    def _median_cut_arrays(colors: np.ndarray, depth: int) -> List[Tuple[int, int, int]]:
        """median_cut on an [n,3] integer array holding the colours in list order: same channel choice (first widest),
        same stable sort, same split, same truncated float mean -- numpy instead of Python lists of tuples."""
        if len(colors) == 0:
            return [(0, 0, 0)]
        if depth == 0:
            n = len(colors)
            sums = colors.sum(axis=0, dtype=np.int64)
            return [tuple(int(int(v) / n) for v in sums)]
        spans = colors.max(axis=0).astype(np.int64) - colors.min(axis=0).astype(np.int64)  # colors: uint8 rows
        ch = int(np.argmax(spans))  # the first of equal spans, like list.index(max(...))
        colors = colors[np.argsort(colors[:, ch], kind="stable")]
        half = len(colors) // 2
        return (ColorReducer._median_cut_arrays(colors[:half], depth - 1) +
                ColorReducer._median_cut_arrays(colors[half:], depth - 1))

    @staticmethod
    def _distinct_in_order(arr: np.ndarray) -> np.ndarray:
        """The distinct rows of an [n,3] uint8 array in order of first occurrence.  Images of 100 000 pixels and more go
        through the device (dp_distinct_first_u8: a 2^24-entry first-index table filled by atomic minima, the first
        occurrences compacted in pixel order; pinned staging buffer up, only the distinct colours down); smaller ones are
        host plumbing (numpy: the PCIe round trip would cost more than the work)."""
        n = len(arr)
        try:
            import torch
            use_gpu = n >= 100_000 and torch.cuda.is_available()
        except ImportError:
            use_gpu = False
        if use_gpu:
            from . import backend
            h_in, _ = _pinned_pair(3 * n)
            np.copyto(h_in.numpy()[:3 * n].reshape(n, 3), arr)   # (arr may be a read-only view of PIL's bytes)
            t = h_in[:3 * n].cuda(non_blocking=True).view(n, 3)
            return backend.distinct_first(t).cpu().numpy()
        packed = (arr[:, 0].astype(np.uint32) << 16) | (arr[:, 1].astype(np.uint32) << 8) | arr[:, 2].astype(np.uint32)
        _, first = np.unique(packed, return_index=True)
        return arr[np.sort(first)]

    @staticmethod
    def _distinct_of_image(rgb):
        """_distinct_in_order for a big PIL 'RGB' image without packing it on the host first: Pillow's four-bytes-per-pixel rows go
        straight into the pinned buffer (_pil_rgbx_into: 2.0 ms for a 4K image where tobytes() + the copy into the buffer took
        5.4), the fourth byte is dropped on the GPU.  None: not applicable (small image, no GPU, no raw encoder) -- the caller packs."""
        n = rgb.size[0] * rgb.size[1]
        try:
            import torch
            if n < 100_000 or not torch.cuda.is_available():
                return None
        except ImportError:
            return None
        from . import backend
        h_in, _ = _pinned_pair(4 * n)
        if not _pil_rgbx_into(rgb, h_in.numpy()[:4 * n]):
            return None
        t = h_in[:4 * n].view(n, 4).cuda(non_blocking=True)[:, :3].contiguous()
        return backend.distinct_first(t).cpu().numpy()

    _replay_ok = None  # does dp_pyset_order_host reproduce THIS interpreter's set order?  (checked once per process)

    @staticmethod
    def _pyset_replay_ok() -> bool:
        """The native median cut replays CPython's set (tuple hash, probing, growth) to obtain the order in which the
        reference's `list(set(image.getdata()))` yields the colours.  That is an implementation detail of the interpreter:
        before relying on it, compare the replay with a real set of this interpreter once -- 70 000 colours, enough to
        cross every growth step incl. the change of policy at 50 000 entries; on any difference (another Python
        implementation or a future CPython) reduce_colors keeps building real sets."""
        if ColorReducer._replay_ok is None:
            ok = False
            try:
                import ctypes as C
                from . import _lib
                L = _lib.load()
                rs = np.random.RandomState(20240607)
                probe = np.ascontiguousarray(rs.randint(0, 256, (70000, 3)).astype(np.uint8))
                probe[:5000] = probe[5000:10000] // 16          # plenty of duplicates and small values as well
                order = np.empty(len(probe), np.uint32)
                nd = C.c_int64(0)
                if L.dp_pyset_order_host(probe.ctypes.data, len(probe), order.ctypes.data, C.byref(nd)) == 0:
                    real = list(set(zip(probe[:, 0].tolist(), probe[:, 1].tolist(), probe[:, 2].tolist())))
                    mine = probe[order[:nd.value]]
                    ok = nd.value == len(real) and np.array_equal(mine, np.array(real, np.uint8).reshape(-1, 3))
            except Exception:  # noqa: BLE001 - library not built, or anything else: the Python path is always right
                ok = False
            ColorReducer._replay_ok = ok
        return ColorReducer._replay_ok

    @staticmethod
    def reduce_colors(image, num_colors: int) -> List[Tuple[int, int, int]]:
        """Median cut over the image's unique colours; returns 2**int(log2(n)) entries
        (dithering_lib.py:1835-1843).  Host-side, and bit-identical to the reference's
        `median_cut(list(set(image.getdata())), depth)`: the stable sort makes the iteration order of that Python set
        observable.  The distinct colours are taken in order of first occurrence (on the GPU for big images; adding a
        colour that is already in a set changes nothing, so the set ends up in the same state); then
        dp_median_cut_host replays CPython's set to get its iteration order and cuts with counting sorts -- a 4K
        photograph with 1.2 M distinct colours: ~0.1 s instead of ~1.5 s with a real set and numpy sorts (reference: ~9 s).
        Should the replay not match this interpreter's sets (_pyset_replay_ok), the set is built by Python as before."""
        rgb = image if image.mode == "RGB" else image.convert("RGB")   # (no copy of an image that is RGB already)
        distinct = ColorReducer._distinct_of_image(rgb)
        if distinct is None:
            distinct = ColorReducer._distinct_in_order(np.frombuffer(rgb.tobytes(), dtype=np.uint8).reshape(-1, 3))
        distinct = np.ascontiguousarray(distinct)
        n = max(int(num_colors), 1)
        depth = int(math.log2(n)) if n > 1 else 0
        if depth <= 10 and ColorReducer._pyset_replay_ok():
            import ctypes as C
            from . import _lib
            out = np.zeros((1 << depth, 3), np.int32)
            n_out = C.c_int(0)
            _lib.check(_lib.load().dp_median_cut_host(distinct.ctypes.data, len(distinct), depth, out.ctypes.data, C.byref(n_out)))
            return [tuple(int(v) for v in c) for c in out[:n_out.value]]
        import itertools
        unique = list(set(zip(distinct[:, 0].tolist(), distinct[:, 1].tolist(), distinct[:, 2].tolist())))
        colors = np.fromiter(itertools.chain.from_iterable(unique), dtype=np.uint8, count=3 * len(unique)).reshape(-1, 3)
        return ColorReducer._median_cut_arrays(colors, depth)

    @staticmethod
    def generate_kmeans_palette(img, num_colors: int, random_state=42) -> List[Tuple[int, int, int]]:
        """k-means palette (dithering_lib.py:1845-1857).  Images of at most 10 000 pixels, where the reference is
        deterministic: the reference's palette (sklearn's seeds, sklearn's labelling of equidistant pixels; 11 reference
        fixtures) -- except that a cluster mean that is an exact integer comes out as that integer, where the reference
        returns it or one less depending on its thread scheduling.  Larger images: the reference fits on an UNSEEDED random
        sample of 10 000 pixels and is not reproducible; here Lloyd runs on the GPU over every pixel (over the colour
        histogram) with exact integer sums, seeded from `random_state` (dither_pie_amd/kmeans.py: parity definition)."""
        from .kmeans import kmeans_palette_from_image
        return kmeans_palette_from_image(img, num_colors, random_state)

    @staticmethod
    def generate_uniform_palette(num_colors: int) -> List[Tuple[int, int, int]]:
        """dithering_lib.py:1859-1872: the first n points of a ceil(n^(1/3))^3 grid, r slowest."""
        side = int(math.ceil(num_colors ** (1 / 3)))
        if side <= 1:
            return [(128, 128, 128)][:num_colors]
        levels = [int(v * 255 / (side - 1)) for v in range(side)]
        grid = [(r, g, b) for r in levels for g in levels for b in levels]
        return grid[:num_colors]


# ------------------------------------------------------------------------------------- ImageDitherer
class ImageDitherer:
    """dithering_lib.py:1877-1992.  Plain attributes only, so instances pickle like the reference's
    (video_processor.py:312-322 ships them to worker processes)."""

    _STRATEGIES = {
        DitherMode.NONE: NoDitherStrategy,
        DitherMode.BAYER: BayerDitherStrategy,
        DitherMode.BLUE_NOISE: BlueNoiseDitherStrategy,
        DitherMode.INTERLEAVED_GRADIENT_NOISE: InterleavedGradientNoiseDitherStrategy,
        DitherMode.ERROR_DIFFUSION: ErrorDiffusionDitherStrategy,
        DitherMode.POLKA_DOT: PolkaDotDitherStrategy,
        DitherMode.PERCEPTUAL: PerceptualDitherStrategy,
        DitherMode.HYBRID: HybridDitherStrategy,
        DitherMode.ADAPTIVE_VARIANCE: AdaptiveVarianceDitherStrategy,
        DitherMode.OSTROMOUKHOV: OstromoukhovDitherStrategy,
    }

    def __init__(self, num_colors: int = 16, dither_mode: Optional[DitherMode] = DitherMode.BAYER,
                 palette: Optional[List[Tuple[int, int, int]]] = None, use_gamma: bool = False,
                 dither_params: Optional[Dict[str, Any]] = None):
        self.num_colors = num_colors
        self.dither_mode = dither_mode
        self.palette = palette
        self.use_gamma = use_gamma
        self.dither_params = dither_params or {}

    @staticmethod
    def get_mode_parameters(mode: DitherMode) -> Optional[Dict[str, Any]]:
        """Parameter metadata for the GUI/CLI (dithering_lib.py:1893-1911); None for modes without
        parameters and for the modes this backend does not implement."""
        cls = ImageDitherer._STRATEGIES.get(mode)
        if cls is None or cls in (NoDitherStrategy, PerceptualDitherStrategy):
            return None  # the reference lists no parameters for these modes either
        return cls.get_parameter_info()

    @staticmethod
    def mode_has_parameters(mode: DitherMode) -> bool:
        return ImageDitherer.get_mode_parameters(mode) is not None

    def _get_dither_strategy(self, mode: DitherMode) -> BaseDitherStrategy:
        """Defaults from get_parameter_info() overlaid by dither_params (dithering_lib.py:1918-1950);
        unknown mode -> ValueError, unknown parameter -> TypeError from the constructor."""
        if mode in _OUT_OF_SCOPE:
            raise NotImplementedError(
                f"dither mode {mode.value!r} is outside the MI355X backend's scope "
                "(none, bayer, blue_noise, IGN, polka_dot, error_diffusion, perceptual, hybrid, "
                "adaptive_variance, ostromoukhov)")
        cls = self._STRATEGIES.get(mode)
        if cls is None:
            raise ValueError(f"Unrecognized DitherMode: {mode}")
        info = cls.get_parameter_info()
        if not info:
            return cls()
        settings = {name: meta["default"] for name, meta in info.items()}
        settings.update(self.dither_params)
        return cls(**settings)

    # -- host plumbing ------------------------------------------------------------------------
    def _ensure_palette(self, first_frame_u8: np.ndarray):
        """palette=None: median cut of the (linearised, when gamma is on) image, kept on the object
        (dithering_lib.py:1960-1966)."""
        if self.palette is None:
            from PIL import Image
            src = _tables.LUT_IN[first_frame_u8] if self.use_gamma else first_frame_u8
            self.palette = ColorReducer.reduce_colors(Image.fromarray(src, "RGB"), self.num_colors)

    def apply_dithering_frames(self, frames, y0: int = 0, x0: int = 0, out=None):
        """uint8 CUDA tensor [N,H,W,3] (or [H,W,3]) -> dithered uint8 CUDA tensor of the same shape.
        Frames stay in HBM; (y0, x0) are the global coordinates of each frame's first pixel when the
        frames are row bands / tiles of a larger image (ordered modes only)."""
        if not self.dither_mode:
            self.dither_mode = DitherMode.NONE
        if self.palette is None:
            first = frames if frames.dim() == 3 else frames[0]
            self._ensure_palette(first.cpu().numpy())
        strategy = self._get_dither_strategy(self.dither_mode)
        import torch
        with torch.cuda.device(frames.device):  # palette, thresholds and launches on the device that holds the frames
            pal = _device_palette(*prepare_palette(self.palette, self.use_gamma))
            return strategy._run(frames, pal, y0=y0, x0=x0, out=out)

    def prepare(self, device=None, accel=True):
        """Create the device-side palette now (and, with accel=True, its search accelerator: ~3.5 ms once) instead of on
        first use / once enough pixels have been served -- for long-running jobs (a video) that know what is coming.
        An addition to the reference's interface; nothing is stored on the (picklable) object."""
        import torch
        if self.palette is None:
            raise ValueError("prepare() needs an explicit palette")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        with torch.cuda.device(dev):
            pal = _device_palette(*prepare_palette(self.palette, self.use_gamma))
            if accel:
                pal.build_accel()
        return self

    def apply_dithering(self, image):
        """PIL image -> PIL 'RGB' image (dithering_lib.py:1952-1992).  Host <-> device copies go through per-thread pinned
        staging buffers that are reused from call to call (the GUI calls this from worker threads).

        Pillow stores an 'RGB' image as FOUR bytes per pixel; packing it to three (tobytes: encode in 64 KB pieces, join them, and
        here one more copy into the pinned buffer) was the largest part of a call (7.6 ms for a 4K image with a 0.02 ms kernel).  So the host side stays in Pillow's own layout: the raw 'RGBX' encoder -- a row memcpy -- writes straight
        into the pinned buffer (2.0 ms instead of 5.4) and the fourth byte is dropped on the GPU; the result comes back packed
        and is unpacked by Image.fromarray (2.5 ms; see below why not mapped).  tools/bench_scripts/pil_probe2.py, pil_probe3.py.
        If Pillow's encoder interface is not what it has been for a decade, the packed path of rounds 1-4 runs instead."""
        import torch
        from PIL import Image
        rgb = image if image.mode == "RGB" else image.convert("RGB")
        w, h = rgb.size
        if self.palette is None:
            self._ensure_palette(np.asarray(rgb))
        n4 = h * w * 4
        pin_in, pin_out = _pinned_pair(n4)
        if w > 0 and h > 0 and _pil_rgbx_into(rgb, pin_in.numpy()):
            dev4 = pin_in.view(h, w, 4).cuda(non_blocking=True)
            dev_out = self.apply_dithering_frames(dev4[..., :3].contiguous())
            out3 = pin_out[:h * w * 3]
            out3.view(h, w, 3).copy_(dev_out.view(h, w, 3), non_blocking=True)
            torch.cuda.current_stream().synchronize()
            # The way back stays packed: an image mapped onto a four-byte buffer carries the raw mode's name ('RGBX'), and
            # convert('RGB') from it is a per-pixel loop (3.5 ms) -- slower than unpacking packed RGB into storage of the image's
            # own (2.5 ms), which also makes the image own its pixels (the staging buffer is overwritten by the next call;
            # tests/test_cabi_and_host.py checks that).  tools/bench_scripts/pil_probe3.py
            return Image.fromarray(out3.numpy().reshape(h, w, 3), "RGB")
        pin_in, pin_out = _pinned_pair(h * w * 3)
        host_in = pin_in.numpy().reshape(h, w, 3)
        np.copyto(host_in.reshape(-1), np.frombuffer(rgb.tobytes(), dtype=np.uint8))
        dev_in = pin_in.view(h, w, 3).cuda(non_blocking=True)
        dev_out = self.apply_dithering_frames(dev_in)
        pin_out.view(h, w, 3).copy_(dev_out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return Image.fromarray(pin_out.numpy().reshape(h, w, 3), "RGB")
