"""Sharding of the hot path over the GPUs of one node (one process per GPU, torch.distributed).

Frames of a video and row bands of one oversize image are independent given global pixel coordinates,
so they shard with no data-path collective; only k-means exchanges data (dither_pie_amd/kmeans.py).
"""
from __future__ import annotations


def shard_range(n: int, rank: int, world: int):
    """Contiguous block [lo, hi) of n items for `rank` of `world` (sizes differ by at most one)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return rank * n // world, (rank + 1) * n // world


def row_bands(h: int, world: int):
    """[(y_lo, y_hi)] row bands of an h-row image, one per rank (8 x 540 rows at 8K)."""
    return [shard_range(h, r, world) for r in range(world)]


def world_info(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def dither_frames_sharded(ditherer, frames_local, out=None):
    """Each rank dithers the frames it holds; nothing is exchanged (video frames are independent)."""
    return ditherer.apply_dithering_frames(frames_local, out=out)


def my_frame_block(n_frames: int, group=None):
    """[lo, hi) of this rank's contiguous block of an n_frames video (one process per GPU)."""
    rank, world = world_info(group)
    return shard_range(n_frames, rank, world)


def gather_frames(local_out, n_frames: int, group=None):
    """All-gather the per-rank blocks of equally shaped frames back into [n_frames, H, W, 3] on every rank
    (only for callers that need the whole video in one place; the dither itself needs no exchange)."""
    import torch
    import torch.distributed as dist
    rank, world = world_info(group)
    if world == 1:
        return local_out
    sizes = [hi - lo for lo, hi in (shard_range(n_frames, r, world) for r in range(world))]
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(local_out.shape[1:]), dtype=local_out.dtype, device=local_out.device)
    buf[: local_out.shape[0]] = local_out
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def visible_devices():
    import torch
    return list(range(torch.cuda.device_count()))


_dev_streams = {}   # (device index, slot) -> torch.cuda.Stream: worker streams are made once per process, not per call


def _worker_stream(dev_index: int, slot: int):
    import torch
    key = (dev_index, slot)
    st = _dev_streams.get(key)
    if st is None:
        st = _dev_streams[key] = torch.cuda.Stream(torch.device("cuda", dev_index))
    return st


def process_on_devices(frames_host, fn, devices=None, chunk: int = 32, out=None):
    """In-process data parallelism over the GPUs of the node -- what the reference's multiprocessing.Pool over frames
    (video_processor.py:304-346) becomes here: the frames [N,H,W,3] (host uint8 tensor, ideally pinned) are cut into one
    contiguous block per device; one worker thread per device copies its block up in chunks on a stream of its own,
    runs `fn(frames_on_device) -> frames_on_device` and copies the result back.  Returns the host tensor
    [N,H',W',3].  Frames are independent: no exchange between devices.  `devices` may name a device more than once
    (two streams on one GPU).  `out`: a host tensor [N,H',W',3] (pinned, kept by the caller between calls) the workers
    write into; without it ONE pinned result tensor is allocated per call, by the first worker that knows H' x W'."""
    import threading
    import torch
    devices = visible_devices() if devices is None else list(devices)
    if not devices:
        raise RuntimeError("no HIP device visible: the MI355X backend has no CPU fallback")
    n = int(frames_host.shape[0])
    blocks = [shard_range(n, i, len(devices)) for i in range(len(devices))]
    errors = []
    res = {"out": out}
    res_mu = threading.Lock()

    def result_for(y):
        with res_mu:
            o = res["out"]
            if o is None:
                o = res["out"] = torch.empty((n,) + tuple(y.shape[1:]), dtype=y.dtype, pin_memory=True)
            elif tuple(o.shape) != (n,) + tuple(y.shape[1:]) or o.dtype != y.dtype:
                raise ValueError(f"out is {tuple(o.shape)} {o.dtype}, the result {(n,) + tuple(y.shape[1:])} {y.dtype}")
            return o

    def work(i, dev_index, lo, hi):
        try:
            dev = torch.device("cuda", dev_index)
            st = _worker_stream(dev_index, i)
            with torch.cuda.device(dev), torch.cuda.stream(st):
                for a in range(lo, hi, chunk):
                    b = min(hi, a + chunk)
                    y = fn(frames_host[a:b].to(dev, non_blocking=True))
                    result_for(y)[a:b].copy_(y, non_blocking=True)
                st.synchronize()
        except Exception as e:  # noqa: BLE001 - reported to the caller below
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i, d, lo, hi)) for i, (d, (lo, hi)) in enumerate(zip(devices, blocks)) if hi > lo]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    if res["out"] is None:
        return torch.empty((0,) + tuple(frames_host.shape[1:]), dtype=torch.uint8)
    return res["out"]


def dither_band(ditherer, band, y_lo: int):
    """Dither a row band of a larger image; `y_lo` is the band's first row in the full image, so the
    threshold tile / IGN field are addressed with global coordinates.  Ordered modes only: an error-diffusion scan
    carries its state across every band boundary (dithering_lib.py:680-687) and does not shard within one image."""
    from .dithering_lib import ORDERED_MODES
    mode = getattr(ditherer, "dither_mode", None)
    if mode is not None and mode not in ORDERED_MODES:
        raise ValueError(f"dither_band: mode {getattr(mode, 'value', mode)!r} does not shard "
                         "within one image (error diffusion crosses every band boundary); ordered modes only")
    return ditherer.apply_dithering_frames(band, y0=y_lo, x0=0)


def gather_bands(band, h: int, group=None):
    """All-gather equal-width uint8 bands back into the full [h, w, 3] image on every rank."""
    import torch
    import torch.distributed as dist
    rank, world = world_info(group)
    if world == 1:
        return band
    w = band.shape[1]
    sizes = [hi - lo for lo, hi in row_bands(h, world)]
    pad = max(sizes)
    buf = torch.zeros((pad, w, 3), dtype=torch.uint8, device=band.device)
    buf[: band.shape[0]] = band
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)
