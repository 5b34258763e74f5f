"""Thin tensor-level wrappers over the C ABI: frames live in torch uint8 tensors on the GPU
(PyTorch-ROCm is used for device memory and streams only), kernels come from
libditherpie_hip.so.  Nothing here computes pixels on the host."""
from __future__ import annotations

import ctypes as C
import threading
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from ._lib import MODE_IGN, MODE_MATRIX, MODE_NEAREST, DitherPieError, check


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    if not torch.cuda.is_available():
        raise DitherPieError(-2, "no HIP device visible: the MI355X backend has no CPU fallback")


class Palette:
    """dp_palette: scipy-order KD-tree + colour tables resident in HBM.

    pal_f32 [K,3] float32 as the KD-tree sees it; out_colors [K,3] uint8 written for each entry;
    lut_in optional 256-entry uint8 map applied to the input bytes."""

    # Building the search accelerator (KD-tree aside: a scan of all 2^24 colours, cell table, tie codes) takes ~3.5 ms up to 256
    # colours and ~25 ms at 1024 (rounds 1-4: 7-12 ms; tools/bench_scripts/accel_build_stages.py), once per palette; the brute-force
    # kernels need ~9 K vector instructions per pixel against ~70 with it.  A palette therefore runs on the brute-force kernels
    # until the pixels it has served would have paid for the build (break-even ~ 1.4e10 / K pixels up to 256 colours: 56 Mpixel
    # = seven 4K frames at 256, 0.9 Gpixel at 16) and builds it then -- never more than twice the cost of the better choice,
    # whatever follows.  One image or a GUI preview never builds it; a video does within its first frames.  build_accel() forces
    # it (long-running jobs that know what is coming).
    ACCEL_BUILD_SECONDS = 0.0035
    ACCEL_BUILD_SECONDS_LARGE = 0.025            # more than 256 colours (128-byte membership masks, deeper tables)
    BRUTE_SECONDS_PER_PIXEL_PER_COLOUR = 2.46e-13

    def accel_break_even_pixels(self):
        build = self.ACCEL_BUILD_SECONDS if self.K <= 256 else self.ACCEL_BUILD_SECONDS_LARGE
        return build / (self.BRUTE_SECONDS_PER_PIXEL_PER_COLOUR * max(self.K, 1))

    def note_pixels(self, n_px):
        """Account n_px pixels about to be processed; builds the accelerator once they have paid for it."""
        if self._accel_done:
            return
        self._px_served += int(n_px)
        if self._px_served >= self.accel_break_even_pixels():
            self.build_accel()

    def __init__(self, pal_f32, out_colors, lut_in=None, accel=False):
        require_gpu()
        self.pal_f32 = np.ascontiguousarray(pal_f32, dtype=np.float32).reshape(-1, 3)
        self.out_colors = np.ascontiguousarray(out_colors, dtype=np.uint8).reshape(-1, 3)
        if self.pal_f32.shape != self.out_colors.shape:
            raise ValueError("pal_f32 and out_colors must both be [K,3]")
        self.lut_in = None if lut_in is None else np.ascontiguousarray(lut_in, dtype=np.uint8)
        if self.lut_in is not None and self.lut_in.size != 256:
            raise ValueError("lut_in must have 256 entries")
        self.K = self.pal_f32.shape[0]
        self._h = C.c_void_p()
        self._destroy = _lib.load().dp_palette_destroy
        check(_lib.load().dp_palette_create(_np_ptr(self.pal_f32), _np_ptr(self.out_colors), self.K,
                                            _np_ptr(self.lut_in), C.byref(self._h)))
        k, integer, nodes = C.c_int(), C.c_int(), C.c_int()
        check(_lib.load().dp_palette_info(self._h, C.byref(k), C.byref(integer), C.byref(nodes)))
        self.is_integer = bool(integer.value)
        self.n_nodes = nodes.value
        self.accel_entries = self.accel_max_list = 0
        self._accel_done = False
        self._px_served = 0
        self._accel_lock = threading.Lock()
        self.device = torch.device("cuda", torch.cuda.current_device())
        if accel:
            self.build_accel()

    def build_accel(self):
        """Build the LDS cell table + tie codes (synchronous, idempotent; a no-op for palettes that do
        not qualify).  Thread-safe: concurrent builders wait for the one build; threads that are launching with the
        palette meanwhile are safe because the library builds into a private copy of the device record and publishes it
        with one assignment, and every launch works on a snapshot of that record (host.cpp: snapshot / publish) - such a
        launch simply still runs without the accelerator."""
        if self._accel_done:
            return
        with self._accel_lock:
            if self._accel_done:
                return
            with torch.cuda.device(self.device):
                check(_lib.load().dp_palette_build_accel(self._h))
            pe, mc = C.c_int(), C.c_int()
            check(_lib.load().dp_palette_accel_info(self._h, C.byref(pe), C.byref(mc)))
            self.accel_entries, self.accel_max_list = pe.value, mc.value
            self._accel_done = True

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._destroy(h)
            self._h = None


class Thresholds:
    """dp_thresholds: a threshold matrix resident in HBM."""

    def __init__(self, handle):
        self._h = handle
        self._destroy = _lib.load().dp_thresholds_destroy
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(_lib.load().dp_thresholds_shape(self._h, C.byref(a), C.byref(b), C.byref(c)))
        self.shape = (a.value, b.value)
        self.integer_form = bool(c.value)

    @classmethod
    def from_matrix(cls, m):
        require_gpu()
        m = np.ascontiguousarray(m, dtype=np.float32)
        if m.ndim != 2:
            raise ValueError("threshold matrix must be 2-D")
        h = C.c_void_p()
        check(_lib.load().dp_thresholds_create(_np_ptr(m), m.shape[0], m.shape[1], C.byref(h)))
        return cls(h)

    @classmethod
    def blue_noise(cls, size, seed):
        require_gpu()
        h = C.c_void_p()
        check(_lib.load().dp_thresholds_blue_noise(int(size), int(seed) & 0xFFFFFFFF, _stream(), C.byref(h)))
        return cls(h)

    def numpy(self):
        out = np.empty(self.shape, np.float32)
        check(_lib.load().dp_thresholds_download(self._h, _np_ptr(out)))
        return out

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._destroy(h)
            self._h = None


def _check_palette_device(pal, f):
    if getattr(pal, "device", f.device) != f.device:
        raise ValueError(f"palette lives on {pal.device}, frames on {f.device}")


def _frames(t):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8):
        raise TypeError("frames must be a CUDA uint8 tensor")
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.dim() != 4 or t.shape[-1] != 3:
        raise ValueError("frames must be [N,H,W,3] or [H,W,3]")
    return t.contiguous()


def _check_out(out, f):
    """A caller-supplied output buffer must be exactly what the kernel writes: uint8, contiguous, same device,
    as many elements as the frames.  Returns it viewed in the frames' [N,H,W,3] shape."""
    if out is None:
        return torch.empty_like(f)
    if not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == torch.uint8):
        raise TypeError("out must be a CUDA uint8 tensor")
    if out.device != f.device:
        raise ValueError(f"out is on {out.device}, frames on {f.device}")
    if out.numel() != f.numel() or not out.is_contiguous():
        raise ValueError("out must be contiguous and hold exactly as many bytes as the frames")
    return out.view(f.shape)


# Scratch per (device, stream): the launch sequences of one call (memset, pass 1, fix-up; progress words, wavefront
# kernel) use it in stream order, so two host threads on the SAME stream must not interleave their sequences (ctypes
# drops the GIL): workspace acquisition + launches run under the lock of that (device, stream).  Different streams
# have different workspaces and run concurrently.
_WS_KEEP_MIN = 16            # (device, stream) scratch buffers kept at least; see _ws_keep()
_WS_SHRINK = 8               # a buffer more than this many times larger than a request is replaced by a smaller one
_ws_cache = OrderedDict()    # key -> [tensor, lock, users]: users = launches inside or waiting for the entry
_ws_guard = threading.Lock()


def _ws_keep():
    """How many (device, stream) scratch buffers are kept before the least recently used idle one is dropped: every device
    with its default stream, its worker stream (sharding.process_on_devices) and its pipe stream (video_processor), plus
    slack -- 8 GPUs in one process are 24 live keys, and a fixed 8 evicted and re-allocated one per batch."""
    return max(_WS_KEEP_MIN, 3 * torch.cuda.device_count() + 4)


class _Launch:
    """with _Launch(device, nbytes) as ws: ... -- the (device, stream) lock held, ws a scratch tensor of >= nbytes."""

    def __init__(self, device, nbytes):
        self.key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        self.device, self.nbytes = device, max(int(nbytes), 1)

    def __enter__(self):
        with _ws_guard:
            ent = _ws_cache.get(self.key)
            if ent is None:
                ent = _ws_cache[self.key] = [None, threading.RLock(), 0]
            ent[2] += 1
            _ws_cache.move_to_end(self.key)
            if len(_ws_cache) > _ws_keep():
                # least recently used first; an entry somebody is inside of (or queued for) is never dropped: its lock is
                # what serialises that stream's launch sequences, a fresh entry for the same key would hand out a second one
                for k in [k for k, e in _ws_cache.items() if e[2] == 0]:
                    if len(_ws_cache) <= _ws_keep():
                        break
                    del _ws_cache[k]
        self.ent = ent
        ent[1].acquire()
        try:
            t = ent[0]
            if t is None or t.numel() < self.nbytes or t.numel() > _WS_SHRINK * max(self.nbytes, 1 << 20):
                t = ent[0] = torch.empty(max(self.nbytes, 1 << 20), dtype=torch.uint8, device=self.device)
        except BaseException:   # out of memory on a grow: __exit__ will not run, the stream's lock must not stay held
            ent[1].release()
            with _ws_guard:
                ent[2] -= 1
            raise
        return t

    def __exit__(self, *exc):
        self.ent[1].release()
        with _ws_guard:
            self.ent[2] -= 1
        return False


def release_workspaces():
    """Drop every cached scratch buffer nobody is using (they are re-created on demand)."""
    with _ws_guard:
        for k in [k for k, e in _ws_cache.items() if e[2] == 0]:
            del _ws_cache[k]


def ordered(frames, pal: Palette, mode, thr: Thresholds | None = None, ign_scale=1.0, ign_seed=0, y0=0, x0=0,
            out=None):
    """nearest / threshold-matrix / IGN dithering of uint8 frames already in HBM -> uint8 frames."""
    f = _frames(frames)
    n, h, w, _ = f.shape
    out = _check_out(out, f)
    _check_palette_device(pal, f)
    L = _lib.load()
    ws_bytes = L.dp_ordered_workspace_bytes(n, h, w)
    with torch.cuda.device(f.device):
        pal.note_pixels(n * h * w)
        with _Launch(f.device, ws_bytes) as ws:
            check(L.dp_ordered_u8(f.data_ptr(), out.data_ptr(), n, h, w, int(y0), int(x0), pal._h, int(mode),
                                  thr._h if thr is not None else None, float(ign_scale), int(ign_seed),
                                  ws.data_ptr(), ws.numel(), _stream()))
    return out.view(frames.shape)


def error_diffusion(frames, pal: Palette, taps, divisor, serpentine=False, out=None, arithmetic="python"):
    """taps: [(dx, dy, weight)] in the reference's list order.  arithmetic: "python" -- the reference's pure-Python
    branch (dithering_lib.py:655-690: KD-tree nearest, float32 products and sums) -- or "numba" -- its
    _error_diffusion_numba branch (:213-308, typed per numba's unification rule: float64 linear-scan nearest, float64 error,
    float64 products and sums rounded on the store; parity-unpinned, fixtures pending)."""
    f = _frames(frames)
    n, h, w, _ = f.shape
    out = _check_out(out, f)
    _check_palette_device(pal, f)
    dx = np.array([t[0] for t in taps], np.int32)
    dy = np.array([t[1] for t in taps], np.int32)
    L = _lib.load()
    ws_bytes = L.dp_error_diffusion_workspace_bytes(n, h, w)
    if arithmetic not in ("python", "numba"):
        raise ValueError("arithmetic must be 'python' or 'numba'")
    with torch.cuda.device(f.device), _Launch(f.device, ws_bytes) as ws:
        if arithmetic == "numba":
            wts = np.array([t[2] for t in taps], np.float32)  # the reference: np.array([...], dtype=np.float32)
            check(L.dp_error_diffusion_numba_u8(f.data_ptr(), out.data_ptr(), n, h, w, pal._h, _np_ptr(dx), _np_ptr(dy),
                                                _np_ptr(wts), float(divisor), len(taps), 1 if serpentine else 0,
                                                ws.data_ptr(), ws.numel(), _stream()))
        else:
            wq = np.array([t[2] / divisor for t in taps], np.float64).astype(np.float32)
            check(L.dp_error_diffusion_u8(f.data_ptr(), out.data_ptr(), n, h, w, pal._h, _np_ptr(dx), _np_ptr(dy),
                                          _np_ptr(wq), len(taps), 1 if serpentine else 0, ws.data_ptr(), ws.numel(),
                                          _stream()))
    return out.view(frames.shape)


def hybrid_numba(frames, pal: Palette, lum_factor=1.0, col_factor=0.2, out=None):
    """HybridDitherStrategy as the reference's numba branch computes it (_hybrid_numba, dithering_lib.py:1396-1494: clamped
    values, float64 linear-scan nearest, float64 luminance / colour split of the error, float64 pushes rounded on the store)."""
    f = _frames(frames)
    n, h, w, _ = f.shape
    out = _check_out(out, f)
    _check_palette_device(pal, f)
    L = _lib.load()
    ws_bytes = L.dp_error_diffusion_workspace_bytes(n, h, w)
    with torch.cuda.device(f.device), _Launch(f.device, ws_bytes) as ws:
        check(L.dp_hybrid_numba_u8(f.data_ptr(), out.data_ptr(), n, h, w, pal._h, float(lum_factor), float(col_factor),
                                   ws.data_ptr(), ws.numel(), _stream()))
    return out.view(frames.shape)


DIFFUSER_PERCEPTUAL, DIFFUSER_HYBRID, DIFFUSER_ADAPTIVE_VARIANCE, DIFFUSER_OSTROMOUKHOV = 1, 2, 3, 4


def variance_gate(frames, pal: Palette, var_threshold=300.0, window_radius=1):
    """uint8 gate map [N,H,W] of AdaptiveVarianceDitherStrategy (local variance >= threshold)."""
    f = _frames(frames)
    n, h, w, _ = f.shape
    gate = torch.empty((n, h, w), dtype=torch.uint8, device=f.device)
    _check_palette_device(pal, f)
    L = _lib.load()
    ws_bytes = L.dp_variance_gate_workspace_bytes(n, h, w)
    with torch.cuda.device(f.device), _Launch(f.device, ws_bytes) as ws:
        check(L.dp_variance_gate_u8(f.data_ptr(), gate.data_ptr(), n, h, w, pal._h, float(var_threshold),
                                    int(window_radius), ws.data_ptr(), ws.numel(), _stream()))
    return gate


def variable_diffusion(frames, pal: Palette, model, p0=0.0, p1=0.0, serpentine=False, gate=None, coef=None, out=None):
    """Perceptual / hybrid / adaptive-variance / Ostromoukhov diffusion of uint8 frames in HBM."""
    f = _frames(frames)
    n, h, w, _ = f.shape
    out = _check_out(out, f)
    _check_palette_device(pal, f)
    L = _lib.load()
    ws_bytes = L.dp_error_diffusion_workspace_bytes(n, h, w)
    if gate is not None and n * h * w >= (1 << 20):
        # adaptive variance: gated-off (flat) regions query the palette with the pixels themselves; exact ties at integer
        # points are then resolved from the accelerator's tie codes instead of a traversal replay per pixel
        pal.build_accel()
    with torch.cuda.device(f.device), _Launch(f.device, ws_bytes) as ws:
        check(L.dp_variable_diffusion_u8(f.data_ptr(), out.data_ptr(), n, h, w, pal._h, int(model), float(p0), float(p1),
                                         1 if serpentine else 0, gate.data_ptr() if gate is not None else None,
                                         coef.data_ptr() if coef is not None else None, ws.data_ptr(), ws.numel(),
                                         _stream()))
    return out.view(frames.shape)


def ign_thresholds(h, w, scale=1.0, seed=0, y0=0, x0=0, device="cuda"):
    require_gpu()
    out = torch.empty((h, w), dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        check(_lib.load().dp_ign_thresholds(out.data_ptr(), h, w, y0, x0, float(scale), int(seed), _stream()))
    return out


def kmeans_step(px, centers, mean=None):
    """px: CUDA uint8 [...,3]; centers: CUDA float64 [K,3]; mean: CUDA float64 [3] or None (sklearn's tie rule, see
    include/ditherpie_hip.h) -> (sums [K,3], counts [K], sumsq [K]) int64"""
    if not (px.is_cuda and px.dtype == torch.uint8 and px.shape[-1] == 3):
        raise TypeError("px must be a CUDA uint8 tensor [...,3]")
    px = px.contiguous()
    centers = centers.to(device=px.device, dtype=torch.float64).contiguous()
    K = centers.shape[0]
    sums = torch.empty((K, 3), dtype=torch.int64, device=px.device)
    counts = torch.empty((K,), dtype=torch.int64, device=px.device)
    sumsq = torch.empty((K,), dtype=torch.int64, device=px.device)
    with torch.cuda.device(px.device):
        if mean is not None:
            mean = mean.to(device=px.device, dtype=torch.float64).contiguous()
        check(_lib.load().dp_kmeans_step_u8(px.data_ptr(), px.numel() // 3, centers.data_ptr(),
                                            mean.data_ptr() if mean is not None else None, K, sums.data_ptr(),
                                            counts.data_ptr(), sumsq.data_ptr(), _stream()))
    return sums, counts, sumsq


def kmeans_step_into(px, centers, totals, want_sq=True, mean=None):
    """One Lloyd pass written into the planar totals buffer `totals` (int64 [5K]: sums [K,3] | counts [K] | squared
    norms [K]); with want_sq=False the last part is left alone.  No allocation, nothing read back.  mean: float64 [3]
    tensor on the device (already contiguous) or None."""
    K = centers.shape[0]
    base = totals.data_ptr()
    with torch.cuda.device(px.device):
        check(_lib.load().dp_kmeans_step_u8(px.data_ptr(), px.numel() // 3, centers.data_ptr(),
                                            mean.data_ptr() if mean is not None else None, K, base, base + 8 * 3 * K,
                                            (base + 8 * 4 * K) if want_sq else None, _stream()))


def distinct_first(px):
    """The distinct colours of the uint8 RGB pixels `px` (CUDA tensor [...,3]) in order of first occurrence
    (dp_distinct_first_u8) -> uint8 tensor [n_distinct, 3] on the device.  One host synchronisation (the count)."""
    if not (px.is_cuda and px.dtype == torch.uint8 and px.shape[-1] == 3):
        raise TypeError("px must be a CUDA uint8 tensor [...,3]")
    px = px.contiguous()
    n = px.numel() // 3
    out = torch.empty((n, 3), dtype=torch.uint8, device=px.device)
    nd = torch.zeros(1, dtype=torch.int64, device=px.device)
    L = _lib.load()
    with torch.cuda.device(px.device):
        with _Launch(px.device, L.dp_distinct_first_workspace_bytes(n)) as ws:
            check(L.dp_distinct_first_u8(px.data_ptr(), n, out.data_ptr(), nd.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
            k = int(nd.item())   # (synchronises: the workspace may be handed to the next call after this)
    return out[:k]


KMEANS_HIST_MAX_K = 256   # dp_kmeans_hist_step: one thread per centre in the list build


class ColourHistogram:
    """count[colour] of uint8 RGB pixels over all 2^24 colours, resident in HBM (dp_kmeans_hist_*): the pixels of a k-means
    fit are read once into it, every Lloyd pass then runs over the histogram (same labels, same int64 totals)."""

    def __init__(self, px=None, device=None):
        require_gpu()
        dev = px.device if px is not None else torch.device(device or "cuda")
        self.buf = torch.empty(_lib.load().dp_kmeans_hist_bytes(), dtype=torch.uint8, device=dev)
        self.n = 0
        if px is not None:
            self.add(px, accumulate=False)

    def add(self, px, accumulate=True):
        if not (px.is_cuda and px.dtype == torch.uint8 and px.shape[-1] == 3):
            raise TypeError("px must be a CUDA uint8 tensor [...,3]")
        if px.device != self.buf.device:
            raise ValueError(f"histogram lives on {self.buf.device}, pixels on {px.device}")
        px = px.contiguous()
        n = px.numel() // 3
        if (self.n if accumulate else 0) + n >= 1 << 32:
            raise ValueError("a colour histogram holds fewer than 2^32 pixels")
        L = _lib.load()
        with torch.cuda.device(px.device), _Launch(px.device, L.dp_kmeans_hist_workspace_bytes(n)) as ws:
            check(L.dp_kmeans_hist_build_u8(px.data_ptr(), n, self.buf.data_ptr(), 1 if accumulate else 0, ws.data_ptr(), ws.numel(),
                                            _stream()))
        self.n = (self.n if accumulate else 0) + n
        return self

    def overflowed(self) -> bool:
        """True when an accumulating build carried a cell's 32-bit pixel count past 2^32 (the device-side check of
        dp_kmeans_hist_build_u8; add() refuses such totals up front, a direct caller of the C ABI must look)."""
        word = self.buf[(1 << 26) + 4 * 8193:(1 << 26) + 4 * 8194].view(torch.int32)
        return bool(int(word.item()) != 0)

    def step_into(self, centers, totals, want_sq=True, mean=None):
        """One Lloyd pass over the histogram into the planar totals buffer (see kmeans_step_into)."""
        K = centers.shape[0]
        base = totals.data_ptr()
        with torch.cuda.device(self.buf.device):
            check(_lib.load().dp_kmeans_hist_step(self.buf.data_ptr(), centers.data_ptr(), mean.data_ptr() if mean is not None else None,
                                                  K, base, base + 8 * 3 * K, (base + 8 * 4 * K) if want_sq else None, _stream()))

    def iterate(self, centers, totals, prev, status, ticket, tol, max_iter, first, mean=None):
        """One whole Lloyd iteration in one launch (dp_kmeans_hist_iterate: pass + centre update, single device).  `totals`
        (int64 [5K]) and `ticket` (int32 [1]) zero before the first iteration."""
        with torch.cuda.device(self.buf.device):
            check(_lib.load().dp_kmeans_hist_iterate(self.buf.data_ptr(), centers.data_ptr(), mean.data_ptr() if mean is not None else None,
                                                     centers.shape[0], totals.data_ptr(), prev.data_ptr(), status.data_ptr(),
                                                     ticket.data_ptr(), float(tol), int(max_iter), 1 if first else 0, _stream()))

    def step(self, centers, mean=None):
        """-> (sums [K,3], counts [K], sumsq [K]) int64, as kmeans_step"""
        centers = centers.to(device=self.buf.device, dtype=torch.float64).contiguous()
        K = centers.shape[0]
        totals = torch.empty(5 * K, dtype=torch.int64, device=self.buf.device)
        if mean is not None:
            mean = mean.to(device=self.buf.device, dtype=torch.float64).contiguous()
        self.step_into(centers, totals, True, mean)
        return totals[:3 * K].view(K, 3), totals[3 * K:4 * K], totals[4 * K:]


def kmeans_update(totals, centers, prev, status, tol, max_iter):
    """The centre update of one Lloyd iteration on the device (dp_kmeans_update); everything stays in HBM."""
    with torch.cuda.device(centers.device):
        check(_lib.load().dp_kmeans_update(totals.data_ptr(), centers.data_ptr(), prev.data_ptr(), status.data_ptr(),
                                           centers.shape[0], float(tol), int(max_iter), _stream()))


KMEANS_PP_MAX_SAMPLE = 16384  # points dp_kmeans_plusplus_u8 holds in LDS


def kmeans_plusplus(sample, K, first, uniforms):
    """sklearn's k-means++ seeding on the device (dp_kmeans_plusplus_u8): `sample` uint8 [n,3] on the GPU, `first` the
    first centre's index, `uniforms` float64 numpy [(K-1), n_trials] drawn by the caller.  -> (ids int32 [K], centers
    float64 [K,3]) on the device, nothing read back."""
    s = sample.reshape(-1, 3)
    if not s.is_contiguous():
        s = s.contiguous()
    n = s.shape[0]
    u = torch.as_tensor(np.ascontiguousarray(uniforms, dtype=np.float64)).reshape(-1).to(s.device)
    n_trials = int(uniforms.shape[1]) if K > 1 else 1
    if K == 1:
        u = torch.zeros(1, dtype=torch.float64, device=s.device)
    ids = torch.empty(K, dtype=torch.int32, device=s.device)
    centers = torch.empty((K, 3), dtype=torch.float64, device=s.device)
    with torch.cuda.device(s.device):
        check(_lib.load().dp_kmeans_plusplus_u8(s.data_ptr(), n, int(K), int(first), u.data_ptr(), n_trials, ids.data_ptr(),
                                                centers.data_ptr(), _stream()))
    return ids, centers


def resize_nearest(frames, oh, ow):
    f = _frames(frames)
    n, h, w, _ = f.shape
    out = torch.empty((n, oh, ow, 3), dtype=torch.uint8, device=f.device)
    with torch.cuda.device(f.device):
        check(_lib.load().dp_resize_nearest_u8(f.data_ptr(), out.data_ptr(), n, h, w, int(oh), int(ow), _stream()))
    return out if frames.dim() == 4 else out[0]


def profile_enable(on=True):
    check(_lib.load().dp_profile_enable(1 if on else 0))


def profile_read():
    """-> (main kernel ms, fix-up ms, launches) accumulated since the last read (HIP events on the
    launch stream)."""
    a, b, n = C.c_double(), C.c_double(), C.c_int64()
    check(_lib.load().dp_profile_read(C.byref(a), C.byref(b), C.byref(n)))
    return a.value, b.value, n.value
