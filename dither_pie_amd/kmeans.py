"""k-means palette extraction on the GPU (ColorReducer.generate_kmeans_palette,
dithering_lib.py:1845-1857 -> sklearn.cluster.KMeans(n_clusters, random_state=42)).

What is kept from the reference: k-means++ seeding with 2+int(ln K) local trials, Lloyd iterations
until the summed squared centre shift is <= 1e-4 * mean(var(X)) or max_iter=300, float64 centres,
`centers.astype(int)` truncation, a list of K 3-tuples as the result.

What differs, and why: the reference fits on `random.sample(range(N), 10000)` drawn from
Python's unseeded global RNG, so two runs of the reference disagree with each other.  Here
  * Lloyd runs over EVERY pixel (dp_kmeans_step_u8, HBM-read bound) with exact integer per-cluster
    totals, so the fit is deterministic and independent of how pixels are sharded over GPUs;
  * only the seeding looks at a sample: the 10 000 pixels at indices RandomState(seed).randint(0,N,10000)
    (all pixels when N <= 10 000), k-means++ on the host in float64 (setup on 30 KB, like the KD-tree
    build; the data-parallel work is on the device);
  * with a torch.distributed process group every rank holds a band of the pixels; the per-iteration
    exchange is ONE all-reduce(sum) of 4K int64 (1 KB at K=32; 5K in the first iteration) over RCCL/xGMI;
  * the centre update and sklearn's stopping rules run on the device (dp_kmeans_update): iterations are launched
    back to back and the host looks at the status words once per 8 iterations.
Parity, two regimes.  At most 10 000 pixels -- the reference is deterministic there, and the fit
reproduces it: sklearn's k-means++ draws (first centre by choice(n, p=uniform)), sklearn's labelling of pixels that are
exactly equidistant from two centres (the rounding of its float64 expression on mean-centred data, as the x86 / OpenBLAS
configuration recorded in tests/golden/kat.json computes it), same iteration count, centres to 1e-9, the same palette on all
11 reference fixtures.  One class of images has no single reference answer: where a cluster's exact mean is an integer
(flat colours, K >= distinct colours) sklearn's chunked float64 sums land on the integer or 1 ulp below it depending on the
order its threads finish, and `astype(int)` gives colour or colour - 1, differently from run to run (tests/golden kmf_*);
the exact integer sums here always give the colour.  More than 10 000 pixels -- the reference samples with an unseeded RNG:
the bar is SURVEY A.6's quality definition (inertia over the full image), and the result is byte-identical for any number
of ranks.
"""
from __future__ import annotations

import numpy as np

SAMPLE = 10000


def first_center_draw(n, rs):
    """sklearn draws the first centre with `random_state.choice(n_samples, p=sample_weight / sample_weight.sum())`
    (sklearn/cluster/_kmeans.py, _kmeans_plusplus; unit weights in KMeans.fit as dithering_lib.py:1854-1856 calls it):
    ONE random_sample() looked up in the normalised cumulative sum with side='right' - not the masked-rejection
    integers of `choice(n)`, which leave the MT19937 stream somewhere else for every later draw."""
    w = np.ones(n, dtype=np.float64)
    return int(rs.choice(n, p=w / w.sum()))


def kmeans_plusplus(sample_u8, K, rs, return_indices=False):
    """sklearn's _kmeans_plusplus on a small pixel sample; rs: numpy RandomState. -> float64 [K,3]
    (return_indices: also the K sample indices picked)"""
    X = np.asarray(sample_u8, dtype=np.float64).reshape(-1, 3)
    n = X.shape[0]
    n_trials = 2 + int(np.log(K))
    centers = np.empty((K, 3), np.float64)
    ids = np.empty(K, np.int64)
    ids[0] = first_center_draw(n, rs)
    centers[0] = X[ids[0]]
    closest = ((X - centers[0]) ** 2).sum(axis=1)
    pot = closest.sum()
    for c in range(1, K):
        picks = np.searchsorted(np.cumsum(closest), rs.uniform(size=n_trials) * pot)
        np.clip(picks, None, n - 1, out=picks)
        d = ((X[picks][:, None, :] - X[None, :, :]) ** 2).sum(axis=2)
        np.minimum(d, closest, out=d)
        pots = d.sum(axis=1)
        best = int(np.argmin(pots))
        pot, closest = pots[best], d[best]
        centers[c] = X[picks[best]]
        ids[c] = picks[best]
    return (centers, ids) if return_indices else centers


def kmeans_plusplus_device(sample, K, rs):
    """The same seeding on the device that holds `sample` (uint8 tensor [n,3]): identical draws from `rs` (they do not
    depend on the data), one launch of dp_kmeans_plusplus_u8 (one workgroup, the sample in LDS, ~0.2 ms instead of
    ~25 ms of numpy on the host), nothing read back.  All distances, prefix sums and potentials are integers below 2^53,
    so the picks are the host version's.  Samples above the kernel's LDS capacity run the same steps as torch ops.
    -> float64 tensor [K,3] on that device"""
    import torch
    from . import backend
    n = sample.reshape(-1, 3).shape[0]
    n_trials = 2 + int(np.log(K))
    if n <= backend.KMEANS_PP_MAX_SAMPLE and n_trials <= 8 and K <= n:
        first = first_center_draw(n, rs)
        uniforms = np.stack([rs.uniform(size=n_trials) for _ in range(1, K)]) if K > 1 else np.zeros((0, n_trials))
        return backend.kmeans_plusplus(sample, K, first, uniforms)[1]
    X = sample.reshape(-1, 3).to(torch.float64)
    centers = torch.empty((K, 3), dtype=torch.float64, device=X.device)
    centers[0] = X[first_center_draw(n, rs)]
    closest = ((X - centers[0]) ** 2).sum(dim=1)
    pot = closest.sum()
    for c in range(1, K):
        r = torch.from_numpy(rs.uniform(size=n_trials)).to(X.device) * pot
        picks = torch.searchsorted(torch.cumsum(closest, 0), r).clamp_(max=n - 1)
        d = ((X[picks][:, None, :] - X[None, :, :]) ** 2).sum(dim=2)
        d = torch.minimum(d, closest)
        pots = d.sum(dim=1)
        best = torch.argmin(pots)
        pot, closest = pots[best], d[best]
        centers[c] = X[picks[best]]
    return centers


def _all_reduce_totals(totals, group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(totals, op=dist.ReduceOp.SUM, group=group)
    return totals


CHECK_EVERY = 8         # iterations launched back to back between two looks at the status words
CHECK_EVERY_FUSED = 16  # ... when an iteration is one launch that returns at once after convergence (dp_kmeans_hist_iterate)


def _check_centres_in_cube(c):
    """dp_kmeans_step_u8's float32 ranking is valid for centres inside the colour cube only (include/ditherpie_hip.h)."""
    import torch
    if isinstance(c, torch.Tensor):
        mm = torch.stack(torch.aminmax(c)).cpu()   # one reduction, one read-back
        lo, hi = float(mm[0]), float(mm[1])
    else:
        lo, hi = float(c.min()), float(c.max())
    if not (lo >= 0.0 and hi <= 255.0):   # also catches NaN
        raise ValueError(f"k-means centres must lie within [0, 255] per channel (got {lo}..{hi})")


HIST_MIN_PIXELS = 1 << 19   # from here on a fit reads its pixels once into the colour histogram and iterates over that


def lloyd(px, init_centers, max_iter=300, tol=1e-4, group=None, step_fn=None, sklearn_ties=True, histogram=None, fuse=None, hist=None,
          centres_are_data_points=False):
    """Lloyd iterations over the uint8 pixels `px` ([...,3] tensor on the GPU).

    histogram: None -- images of HIST_MIN_PIXELS and more (per rank) with K <= 256 are read ONCE into count[colour]
    (backend.ColourHistogram, dp_kmeans_hist_*) and every pass runs over the histogram: a label is a function of the colour,
    the totals are sums of count x colour, so labels and int64 totals are those of the passes over the pixels, bit for bit,
    at 16 KB per occupied cell of the colour cube per pass instead of 3 B per pixel.  True / False force the choice.
    hist: a backend.ColourHistogram of exactly these pixels that the caller has built already (fit_palette builds it on a second
    stream while the seeding runs).
    fuse: None -- with the histogram and no other rank to exchange totals with, an iteration is ONE launch
    (dp_kmeans_hist_iterate: the pass, and the centre update by the workgroup that finishes last); False keeps the three
    steps pass / all-reduce / update (what a sharded fit runs).

    sklearn_ties: a pass "zero" (one centre: the totals of all pixels, all-reduced like any other pass) gives the data
    mean KMeans.fit subtracts, and every pass labels equidistant pixels as sklearn's float64 expression on the centred
    data does (dp_kmeans_step_u8's mean_dev) - what makes fits of <= 10 000 pixels, where the reference is
    deterministic (dithering_lib.py:1845-1857), equal the reference's also on structured images.  False: lowest index.

    Each iteration is three stream-ordered steps with no host synchronisation: the pass over the pixels
    (dp_kmeans_step_u8: labels + exact integer totals), ONE all-reduce of the planar int64 totals when a process group is
    active (4K words; 5K in the first iteration), and the centre update with sklearn's stopping rules on the device
    (dp_kmeans_update).  The host reads the 8 status words back once per CHECK_EVERY iterations; iterations launched
    past convergence change nothing.  With step_fn (tests on CPU tensors: the oracle's pass standing in for the kernel)
    the same rules run on the host.  Returns (centers float64 [K,3] numpy, inertia, n_iter)."""
    import torch
    if step_fn is not None:
        return _lloyd_host(px, init_centers, max_iter, tol, group, step_fn, sklearn_ties)
    from . import backend
    flat = px.reshape(-1, 3)
    if not flat.is_contiguous():
        flat = flat.contiguous()
    dev = flat.device
    if isinstance(init_centers, torch.Tensor):
        centers = init_centers.to(device=dev, dtype=torch.float64).reshape(-1, 3).contiguous().clone()
    else:
        centers = torch.as_tensor(np.array(init_centers, dtype=np.float64).reshape(-1, 3)).to(dev).contiguous()
    if not centres_are_data_points:   # (k-means++ seeds are pixels of the image: inside the cube by construction, nothing to read back)
        _check_centres_in_cube(centers)
    K = centers.shape[0]
    totals = torch.zeros(5 * max(K, 1), dtype=torch.int64, device=dev)
    prev = torch.zeros(4 * K, dtype=torch.int64, device=dev)
    status = torch.zeros(8, dtype=torch.float64, device=dev)
    if histogram is None:
        histogram = hist is not None or flat.shape[0] >= HIST_MIN_PIXELS
    if hist is not None and (hist.n != flat.shape[0] or K > backend.KMEANS_HIST_MAX_K or histogram is False):
        hist = None
    if hist is None and histogram and K <= backend.KMEANS_HIST_MAX_K and 0 < flat.shape[0] < (1 << 32):
        hist = backend.ColourHistogram(flat)
    if hist is not None:
        def one_pass(c, tot, want_sq, mean):
            hist.step_into(c, tot, want_sq, mean)
    else:
        def one_pass(c, tot, want_sq, mean):
            backend.kmeans_step_into(flat, c, tot, want_sq=want_sq, mean=mean)
    mean = None
    if sklearn_ties:
        t0 = torch.zeros(4, dtype=torch.int64, device=dev)
        one_pass(torch.zeros((1, 3), dtype=torch.float64, device=dev), t0, False, None)
        _all_reduce_totals(t0, group)
        mean = (t0[:3].to(torch.float64) / t0[3].to(torch.float64)).contiguous()   # exact sums: sum / n rounded once
    import torch.distributed as dist
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    fused = hist is not None and not sharded and fuse is not False   # one device: ONE launch per iteration (dp_kmeans_hist_iterate)
    ticket = torch.zeros(1, dtype=torch.int32, device=dev) if fused else None
    launched = 0
    every = CHECK_EVERY_FUSED if fused else CHECK_EVERY
    while True:
        for _ in range(every):
            first = launched == 0
            if fused:
                hist.iterate(centers, totals, prev, status, ticket, tol, max_iter, first, mean)
            else:
                one_pass(centers, totals, first, mean)
                _all_reduce_totals(totals if first else totals[:4 * K], group)
                backend.kmeans_update(totals, centers, prev, status, tol, max_iter)
            launched += 1
        st = status.cpu()
        if int(st[0]) in (1, 3):
            break
        if launched > max_iter + every + 2:  # cannot happen (the update kernel stops at max_iter)
            raise RuntimeError("k-means did not terminate")
    return centers.cpu().numpy(), float(st[2]), int(st[1])


def _lloyd_host(px, init_centers, max_iter, tol, group, step_fn, sklearn_ties=True):
    """The same iteration with the centre update on the host side of `step_fn(px, centers, mean)` (CPU tests of the
    N>1 logic)."""
    import torch
    centers = torch.as_tensor(np.array(init_centers, dtype=np.float64).reshape(-1, 3))
    _check_centres_in_cube(centers)
    K = centers.shape[0]
    dev = None
    tol_abs = None
    prev = None
    inertia = float("nan")
    n_iter = 0
    mean = None
    if sklearn_ties:   # pass zero: one centre, the totals of every pixel over all ranks
        s0, n0, _ = step_fn(px, torch.zeros((1, 3), dtype=torch.float64), None)
        t0 = _all_reduce_totals(torch.cat([s0.reshape(3), n0.reshape(1)]).contiguous(), group)
        mean = (t0[:3].to(torch.float64) / t0[3].to(torch.float64)).contiguous()

    def totals_for(c):
        sums, counts, sumsq = step_fn(px, c, mean)
        t = torch.cat([sums.reshape(K, 3), counts.reshape(K, 1), sumsq.reshape(K, 1)], dim=1).contiguous()
        return _all_reduce_totals(t, group)

    def inertia_of(c, t):
        s, n, q = t[:, :3].double(), t[:, 3].double(), t[:, 4].double()
        return (q - 2.0 * (c * s).sum(1) + n * (c * c).sum(1)).sum()

    for n_iter in range(1, max_iter + 1):
        totals = totals_for(centers)
        if dev is None:
            dev = totals.device
            centers = centers.to(dev)
        s, n = totals[:, :3].double(), totals[:, 3].double()
        if tol_abs is None:  # sklearn: tol * mean of the per-channel variances (from the exact totals)
            N = n.sum()
            mu = s.sum(0) / N
            tol_abs = tol * float(((totals[:, 4].double().sum() / N - (mu * mu).sum()) / 3.0).item())
        new = torch.where((n > 0).unsqueeze(1), s / n.clamp(min=1.0).unsqueeze(1), centers)
        shift = ((new - centers) ** 2).sum()
        same = torch.zeros((), dtype=torch.float64, device=dev) if prev is None else \
            (prev == totals[:, :4]).all().double()
        stats = torch.stack([shift, inertia_of(centers, totals), same]).cpu()
        prev = totals[:, :4].clone()
        inertia = float(stats[1])
        if bool(stats[2] > 0):   # assignments did not change: strict convergence, centres stay
            break
        centers = new
        if float(stats[0]) <= tol_abs or n_iter == max_iter:
            inertia = float(inertia_of(centers, totals_for(centers)).item())  # inertia of the final centres
            break
    return centers.cpu().numpy(), inertia, n_iter


_sample_idx = {}   # (device index, n_total, random_state) -> int64 device tensor; a handful of image sizes per process


def _sample_indices(n_total, random_state, device):
    import torch
    if not isinstance(random_state, (int, np.integer)):   # (None or an array seed: nothing to key on, draw afresh)
        return torch.from_numpy(np.random.RandomState(random_state).randint(0, n_total, SAMPLE).astype(np.int64)).to(device)
    key = (device.index, int(n_total), int(random_state))
    t = _sample_idx.get(key)
    if t is None:
        if len(_sample_idx) >= 16:
            _sample_idx.clear()
        idx = np.random.RandomState(random_state).randint(0, n_total, SAMPLE)
        t = _sample_idx[key] = torch.from_numpy(idx.astype(np.int64)).to(device)
    return t


def seed_sample(px, n_total, offset, random_state, group=None, as_tensor=False):
    """The pixels at the global indices RandomState(random_state).randint(0, n_total, SAMPLE) (all of them
    when n_total <= SAMPLE), gathered from the local band [offset, offset+len(px)) and summed across ranks.
    -> uint8 numpy array [n,3] (as_tensor: uint8 tensor on px's device)"""
    import torch
    import torch.distributed as dist
    flat = px.reshape(-1, 3)
    alone = not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1)
    if alone and offset == 0 and flat.shape[0] == n_total and flat.is_cuda:
        # one rank holding the whole image: one gather with the index tensor kept on the device (the indices depend on the
        # image SIZE and the seed only -- every frame of a video draws the same ones; 0.31 -> 0.05 ms per fit)
        if n_total <= SAMPLE:
            got = flat
        else:
            got = flat.index_select(0, _sample_indices(n_total, random_state, flat.device))
        return got.contiguous() if as_tensor else got.cpu().numpy()
    if n_total <= SAMPLE:
        idx = np.arange(n_total)
    else:
        idx = np.random.RandomState(random_state).randint(0, n_total, SAMPLE)
    local = (idx >= offset) & (idx < offset + flat.shape[0])
    buf = torch.zeros((len(idx), 3), dtype=torch.int64, device=flat.device)
    if local.any():
        sel = torch.from_numpy(idx[local] - offset).to(flat.device)
        buf[torch.from_numpy(np.nonzero(local)[0]).to(flat.device)] = flat[sel].to(torch.int64)
    buf = _all_reduce_totals(buf, group)
    if as_tensor:
        return buf.to(torch.uint8)
    return buf.cpu().numpy().astype(np.uint8)


_side_streams = {}


def _side_stream(device):
    """One extra stream per device (kept: creating a stream per fit costs more than the overlap gains)."""
    import torch
    key = device.index if device.index is not None else torch.cuda.current_device()
    s = _side_streams.get(key)
    if s is None:
        s = _side_streams[key] = torch.cuda.Stream(device=device)
    return s


def fit_palette(px, K, random_state=42, n_total=None, offset=0, group=None, max_iter=300, tol=1e-4):
    """px: uint8 CUDA tensor [...,3] holding this rank's band of the image.  -> (palette list, centers, inertia, n_iter)"""
    n_local = px.numel() // 3
    n_total = n_local if n_total is None else int(n_total)
    hist = None
    if px.is_cuda:
        import torch
        from . import backend
        flat = px.reshape(-1, 3)
        if HIST_MIN_PIXELS <= n_local < (1 << 32) and K <= backend.KMEANS_HIST_MAX_K:
            # the colour histogram does not depend on the seeds: it is built on a second stream while the sample is gathered and the
            # k-means++ kernel (one workgroup) runs
            cur = torch.cuda.current_stream(px.device)
            side = _side_stream(px.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                hist = backend.ColourHistogram(flat if flat.is_contiguous() else flat.contiguous())
        sample = seed_sample(px, n_total, offset, random_state, group, as_tensor=True)
        init = kmeans_plusplus_device(sample, K, np.random.RandomState(random_state))
        if hist is not None:
            cur.wait_stream(side)
            hist.buf.record_stream(cur)
    else:
        sample = seed_sample(px, n_total, offset, random_state, group)
        init = kmeans_plusplus(sample, K, np.random.RandomState(random_state))
    centers, inertia, n_iter = lloyd(px, init, max_iter=max_iter, tol=tol, group=group, hist=hist, centres_are_data_points=True)
    palette = [tuple(int(v) for v in c) for c in centers.astype(int)]
    return palette, centers, inertia, n_iter


def kmeans_palette_from_image(img, K, random_state=42):
    """PIL image -> list of K (r,g,b) int tuples; the whole image goes to the current GPU."""
    import torch
    arr = np.array(img.convert("RGB"))
    px = torch.from_numpy(arr.reshape(-1, 3)).cuda()
    palette, _, _, _ = fit_palette(px, int(K), random_state)
    return palette
