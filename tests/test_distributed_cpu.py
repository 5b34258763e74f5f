"""CPU tier: the N>1 path (k-means exchange, frame/row-band partition) with world_size-2 gloo.

The device kernel is replaced by the oracle's Lloyd pass as `step_fn` (the checker standing in for the
kernel, in tests only); what is under test is the host logic: packing of the integer totals, the single
all-reduce per iteration, the seeding sample gathered from sharded bands, and rank-count independence."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_step(px, centers, mean=None):
    """-> (sums [K,3], counts [K], squared norms [K]); only the TOTAL of the squared norms is ever used (tolerance and
    inertia), so they are booked on cluster 0."""
    import torch
    from oracle import oracle as orc
    p = px.reshape(-1, 3).cpu().numpy()
    c = centers.cpu().numpy()
    sums, counts, _ = orc.kmeans_step(p, c, None if mean is None else mean.cpu().numpy())
    sq = np.zeros(len(c), np.int64)
    x = p.astype(np.int64)
    sq[0] = int((x * x).sum())
    return torch.from_numpy(sums), torch.from_numpy(counts), torch.from_numpy(sq)


def _worker(rank, world, port, h, w, K, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dither_pie_amd import kmeans, sharding
        from oracle import oracle as orc
        img = orc.rnd(h, w, 77)
        lo, hi = sharding.shard_range(h, rank, world)
        band = torch.from_numpy(np.ascontiguousarray(img[lo:hi]))
        sample = kmeans.seed_sample(band, h * w, lo * w, 42)
        init = kmeans.kmeans_plusplus(sample, K, np.random.RandomState(42))
        centers, inertia, n_iter = kmeans.lloyd(band, init, step_fn=_oracle_step)
        q.put((rank, centers, inertia, n_iter, sample))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_kmeans_two_ranks_equal_one_rank(world):
    """world = 8: the shape of the driver's SCALE run (eight row bands, one int64 all-reduce per Lloyd iteration, the seeding
    sample summed from eight bands) over gloo on the CPU -- h = 123 rows do not divide by eight, so the bands differ in height."""
    import torch
    import torch.multiprocessing as mp
    from dither_pie_amd import kmeans
    from oracle import oracle as orc
    h, w, K = (120 if world == 2 else 123), 101, 8   # > 10000 px: the seeding sample is a strict subset
    img = orc.rnd(h, w, 77)
    full = torch.from_numpy(img)
    sample1 = kmeans.seed_sample(full, h * w, 0, 42)
    init = kmeans.kmeans_plusplus(sample1, K, np.random.RandomState(42))
    c1, i1, n1 = kmeans.lloyd(full, init, step_fn=_oracle_step)

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, h, w, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, c, i, n, sample in res:
        assert np.array_equal(sample, sample1)
        assert np.array_equal(c, c1), "centres must be byte-identical for any rank count"
        assert i == i1 and n == n1
    # and the host loop agrees with the oracle's own Lloyd from the same start
    c_or, i_or, _ = orc.kmeans_lloyd(img.reshape(-1, 3), init)
    assert np.abs(c_or - c1).max() < 1e-9 and abs(i_or - i1) <= 1e-9 * i_or


def test_lloyd_matches_sklearn_fixture_from_same_init(gold, kat):
    import torch
    from dither_pie_amd import kmeans
    from oracle import oracle as orc
    m = kat["misc"]["km16"]
    arr = orc.grad(m["h"], m["w"])
    px = arr.reshape(-1, 3)
    init = px[gold["km16_init_idx"]].astype(np.float64)
    centers, inertia, n_iter = kmeans.lloyd(torch.from_numpy(px), init, step_fn=_oracle_step)
    assert np.abs(centers - gold["km16_centers"]).max() < 1e-6
    assert abs(inertia - m["inertia"]) <= 1e-6 * m["inertia"]


class _OracleDitherer:
    """Stand-in for ImageDitherer on CPU tensors (tests only): the oracle does the pixels, so that the partition logic
    of dither_pie_amd.sharding -- who takes which frames / rows, with which global offsets -- is what is under test."""

    def __init__(self, pal, mode, params):
        self.pal, self.mode, self.params = pal, mode, params

    def apply_dithering_frames(self, frames, y0=0, x0=0, out=None):
        import torch
        from oracle import oracle as orc
        a = frames.numpy()
        if a.ndim == 3:
            return torch.from_numpy(orc.apply_dithering(a, self.pal, self.mode, self.params, False, y0=y0, x0=x0))
        if a.shape[0] == 0:    # a rank with no frames (7 frames over 8 ranks): the product returns an empty batch too
            return torch.from_numpy(a.copy())
        return torch.from_numpy(np.stack([orc.apply_dithering(f, self.pal, self.mode, self.params, False, y0=y0, x0=x0) for f in a]))


def _shard_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dither_pie_amd import sharding
        from oracle import oracle as orc
        pal = orc.palr(16, 3)
        # video: 7 frames in contiguous blocks (C5's partition), gathered back in frame order
        frames = np.stack([orc.rnd(24, 40, 100 + i) for i in range(7)])
        dv = _OracleDitherer(pal, "bayer", {"size": "4x4"})
        lo, hi = sharding.my_frame_block(7)
        mine = sharding.dither_frames_sharded(dv, torch.from_numpy(frames[lo:hi]))
        video = sharding.gather_frames(mine, 7)
        # one image in row bands with global coordinates (C4's partition), gathered back
        img = orc.rnd(45, 52, 9)
        db = _OracleDitherer(pal, "blue_noise", {"size": 32, "seed": 0})
        blo, bhi = sharding.shard_range(45, rank, world)
        band = sharding.dither_band(db, torch.from_numpy(np.ascontiguousarray(img[blo:bhi])), blo)
        whole = sharding.gather_bands(band, 45)
        q.put((rank, (lo, hi), video.numpy(), (blo, bhi), whole.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_frames_and_row_bands_two_ranks_equal_one_rank(world):
    """world 8 is the driver's SCALE shape; with 7 frames one of the eight ranks holds an empty block."""
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    pal = orc.palr(16, 3)
    frames = np.stack([orc.rnd(24, 40, 100 + i) for i in range(7)])
    ref_video = np.stack([orc.apply_dithering(f, pal, "bayer", {"size": "4x4"}) for f in frames])
    ref_img = orc.apply_dithering(orc.rnd(45, 52, 9), pal, "blue_noise", {"size": 32, "seed": 0})
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if world == 2:
        assert [r[1] for r in res] == [(0, 3), (3, 7)] and [r[3] for r in res] == [(0, 22), (22, 45)]
    else:   # contiguous, ordered, covering; sizes differ by at most one
        for which, total in ((1, 7), (3, 45)):
            spans = [r[which] for r in res]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    for rank, _, video, _, whole in res:
        assert np.array_equal(video, ref_video), rank
        assert np.array_equal(whole, ref_img), rank


def test_dither_band_is_for_ordered_modes_only():
    """sharding.dither_band says so itself: an error-diffusion scan does not shard within one image."""
    from dither_pie_amd import sharding
    from dither_pie_amd.dithering_lib import DitherMode, ImageDitherer
    d = ImageDitherer(16, DitherMode.ERROR_DIFFUSION, [(0, 0, 0), (255, 255, 255)], False, {})
    with pytest.raises(ValueError, match="ordered modes only"):
        sharding.dither_band(d, None, 10)
