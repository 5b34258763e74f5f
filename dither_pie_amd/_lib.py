"""ctypes binding of libditherpie_hip.so (C ABI: include/ditherpie_hip.h).

The library is the product: there is no CPU fallback.  If the shared object is missing or a
call fails, a DitherPieError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# Two builds of the same sources (csrc/Makefile): libditherpie_hip.so -- the product, which reads no environment variable --
# and libditherpie_hip_exp.so (-DDP_EXPERIMENTS), in which the DP_* switches that force a table / kernel / schedule are
# compiled in.  The product library is what load() returns unless DITHER_PIE_EXPERIMENTS=1 is set (tools/bench_scripts) or
# select(True) was called (the `switches` fixture of tests/conftest.py: only the tests that set a DP_* switch run on the
# twin; every other test, golden and fuzzer runs on the library that ships).
# DP_LIB_PATH: another build altogether (A/B measurements of kernel variants: tools/bench_scripts/ab_kernel.py)
PRODUCT_PATH = os.path.join(_HERE, "libditherpie_hip.so")
EXPERIMENTS_PATH = os.path.join(_HERE, "libditherpie_hip_exp.so")
EXPERIMENTS = os.environ.get("DITHER_PIE_EXPERIMENTS", "") not in ("", "0")
LIB_PATH = os.environ.get("DP_LIB_PATH") or (EXPERIMENTS_PATH if EXPERIMENTS else PRODUCT_PATH)
CSRC = os.path.join(_HERE, "csrc")
# the ABI revision this binding was written against (include/ditherpie_hip.h: DP_ABI_VERSION); load() refuses a library
# that reports another one -- a stale build bound through DP_LIB_PATH would otherwise read K as a pointer
ABI_VERSION = 102

DP_OK, DP_EINVAL, DP_EUNSUPPORTED, DP_EHIP, DP_ENOMEM, DP_EWORKSPACE = range(6)
DP_MAX_COLORS = 1024
MODE_NEAREST, MODE_MATRIX, MODE_IGN = 0, 1, 2


class DitherPieError(RuntimeError):
    """A libditherpie_hip call failed (the message is dp_last_error())."""

    def __init__(self, code, msg):
        super().__init__(f"libditherpie_hip error {code}: {msg}")
        self.code = code


_lock = threading.Lock()
_lib = None
_loaded = {}   # path -> CDLL (both builds may be mapped in one process; their device records have the same layout)

_vp, _i, _i64, _sz, _f = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float
_SIGS = {
    "dp_version": (C.c_int, []),
    "dp_last_error": (C.c_char_p, []),
    "dp_device_info": (_i, [C.POINTER(C.c_int), C.c_char_p, _sz]),
    "dp_palette_create": (_i, [_vp, _vp, _i, _vp, C.POINTER(_vp)]),
    "dp_palette_destroy": (None, [_vp]),
    "dp_palette_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "dp_palette_build_accel": (_i, [_vp]),
    "dp_palette_accel_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "dp_kdtree_build_host": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i)]),
    "dp_thresholds_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "dp_thresholds_blue_noise": (_i, [_i, C.c_uint32, _vp, C.POINTER(_vp)]),
    "dp_thresholds_shape": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "dp_thresholds_download": (_i, [_vp, _vp]),
    "dp_thresholds_destroy": (None, [_vp]),
    "dp_ign_thresholds": (_i, [_vp, _i, _i, _i, _i, _f, _i, _vp]),
    "dp_ordered_workspace_bytes": (_sz, [_i64, _i, _i]),
    "dp_ordered_u8": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp, _i, _vp, _f, _i, _vp, _sz, _vp]),
    "dp_error_diffusion_workspace_bytes": (_sz, [_i64, _i, _i]),
    "dp_error_diffusion_u8": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "dp_error_diffusion_numba_u8": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, C.c_double, _i, _i, _vp, _sz, _vp]),
    "dp_hybrid_numba_u8": (_i, [_vp, _vp, _i64, _i, _i, _vp, C.c_double, C.c_double, _vp, _sz, _vp]),
    "dp_variable_diffusion_u8": (_i, [_vp, _vp, _i64, _i, _i, _vp, _i, _f, _f, _i, _vp, _vp, _vp, _sz, _vp]),
    "dp_variance_gate_workspace_bytes": (_sz, [_i64, _i, _i]),
    "dp_variance_gate_u8": (_i, [_vp, _vp, _i64, _i, _i, _vp, _f, _i, _vp, _sz, _vp]),
    "dp_kmeans_step_u8": (_i, [_vp, _i64, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "dp_kmeans_hist_bytes": (_sz, []),
    "dp_kmeans_hist_workspace_bytes": (_sz, [_i64]),
    "dp_kmeans_hist_build_u8": (_i, [_vp, _i64, _vp, _i, _vp, _sz, _vp]),
    "dp_kmeans_hist_step": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "dp_kmeans_hist_iterate": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, C.c_double, _i, _i, _vp]),
    "dp_kmeans_update": (_i, [_vp, _vp, _vp, _vp, _i, C.c_double, _i, _vp]),
    "dp_kmeans_plusplus_u8": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "dp_resize_nearest_u8": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "dp_distinct_first_workspace_bytes": (_sz, [_i64]),
    "dp_distinct_first_u8": (_i, [_vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "dp_pyset_order_host": (_i, [_vp, _i64, _vp, C.POINTER(_i64)]),
    "dp_median_cut_host": (_i, [_vp, _i64, _i, _vp, C.POINTER(_i)]),
    "dp_profile_enable": (_i, [_i]),
    "dp_profile_read": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i64)]),
}
EXPORTS = tuple(_SIGS)


def build(force=False):
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC])
    return LIB_PATH


def load():
    """Return the loaded library; raises DitherPieError when it is not built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise DitherPieError(-1, f"{LIB_PATH} is missing: build it with "
                                         f"`make -C {CSRC}` (or dither_pie_amd.build()); there is no CPU fallback")
            # The library shares the process's HIP runtime with PyTorch (device memory and streams are torch's).
            # torch ships its own libamdhip64: import it first so that ours binds to that copy -- loading the
            # system runtime before torch's leaves two runtimes in the process and this one without a device.
            try:
                import torch  # noqa: F401
            except ImportError:  # host-only use (dp_kdtree_build_host, symbol checks)
                pass
            L = _loaded.get(LIB_PATH)
            if L is None:
                L = C.CDLL(LIB_PATH)
                # the version first, bound alone: a stale library may lack entry points of _SIGS altogether, and the
                # message a user needs is "rebuild it", not an AttributeError out of the signature loop
                try:
                    ver = L.dp_version
                except AttributeError:
                    raise DitherPieError(-1, f"{LIB_PATH} exports no dp_version: not a libditherpie_hip build; rebuild it "
                                             f"with `make -C {CSRC}`") from None
                ver.restype, ver.argtypes = C.c_int, []
                got = ver()
                if got != ABI_VERSION:
                    raise DitherPieError(-1, f"{LIB_PATH} reports ABI version {got}, this binding was written for "
                                             f"{ABI_VERSION}: rebuild it with `make -C {CSRC}`")
                for name, (res, args) in _SIGS.items():
                    try:
                        fn = getattr(L, name)
                    except AttributeError:
                        raise DitherPieError(-1, f"{LIB_PATH} (ABI {got}) does not export {name}: rebuild it with "
                                                 f"`make -C {CSRC}`") from None
                    fn.restype = res
                    fn.argtypes = args
                _loaded[LIB_PATH] = L
            _lib = L
    return _lib


def select(experiments):
    """Make load() return the product library (False) or its -DDP_EXPERIMENTS twin (True) from now on; returns the previous
    choice.  Device objects (palettes, thresholds) made through one library must not be handed to the other: callers drop
    their caches (dithering_lib.drop_device_caches) around a switch.  Test infrastructure; the product never calls it."""
    global _lib, EXPERIMENTS, LIB_PATH
    with _lock:
        prev = EXPERIMENTS
        EXPERIMENTS = bool(experiments)
        LIB_PATH = EXPERIMENTS_PATH if EXPERIMENTS else PRODUCT_PATH
        _lib = None
    return prev


def check(rc):
    if rc != DP_OK:
        msg = load().dp_last_error()
        raise DitherPieError(rc, msg.decode("utf-8", "replace") if msg else "unknown error")


def device_info():
    n = C.c_int(0)
    buf = C.create_string_buffer(128)
    check(load().dp_device_info(C.byref(n), buf, 128))
    return n.value, buf.value.decode()
