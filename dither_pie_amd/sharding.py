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


def dither_band(ditherer, band, y_lo: int):
    """Dither a row band of a larger image; `y_lo` is the band's first row in the full image, so the
    threshold tile / IGN field are addressed with global coordinates (ordered modes only)."""
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
