"""GPU tier: BASELINE.json's configurations at their FULL sizes, and the robustness paths around the kernels.

Chain of custody of the expected values: `kat.json` holds hashes computed by the REFERENCE itself
(tests/golden/make_golden.py) for C1 (512x512, median-cut 16), C2 (4K, 256 colours), C3 (4K Floyd-Steinberg, ~7 min in
the reference), C4's dither half (7680x4320 blue noise) and one C5 frame; where a test compares with the C oracle
instead (batches of other frames, k-means totals), the oracle is the one tests/test_oracle_golden.py pins against those
same reference hashes at the same sizes."""
import hashlib
import json
import os
import threading

import numpy as np
import pytest

from conftest import GOLDEN, case_input, case_palette

pytestmark = pytest.mark.gpu

with open(os.path.join(GOLDEN, "kat.json")) as _f:
    _KAT = json.load(_f)
_BY_NAME = {c["name"]: c for c in _KAT["cases"]}


def H(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


@pytest.fixture(scope="module")
def d():
    import torch
    assert torch.cuda.is_available()
    from dither_pie_amd import dithering_lib
    return dithering_lib


@pytest.fixture(scope="module")
def be():
    from dither_pie_amd import backend
    return backend


def _need(name):
    if name not in _BY_NAME:
        pytest.skip(f"{name} not in kat.json (regenerate with tests/golden/make_golden.py --append)")
    return _BY_NAME[name]


# ------------------------------------------------------------------------------------------------ C3
def test_c3_floyd_steinberg_4k_both_schedules(d, orc, switches):
    """rnd(2160,3840,1234), U16, Floyd-Steinberg: one frame (bands spread over workgroups), the same frame with one
    workgroup per frame, and a 24-frame batch, all against the reference's own hash of that frame."""
    import torch
    case = _need("c3_ed_fs_U16_rnd4k")
    arr = case_input(orc, case["input"])
    assert H(arr) == case["h_in"]
    pal = case_palette(orc, case["palette"])
    it = d.ImageDitherer(len(pal), d.DitherMode.ERROR_DIFFUSION, pal, False, dict(case["params"]))
    x = torch.from_numpy(arr).cuda()
    assert H(it.apply_dithering_frames(x).cpu().numpy()) == case["h_out"]          # G > 1 schedule
    switches.setenv("DP_ED_ONE_WG", "1")
    assert H(it.apply_dithering_frames(x).cpu().numpy()) == case["h_out"]          # one workgroup per frame
    switches.delenv("DP_ED_ONE_WG")
    # a batch of 24: frames 0, 7, 23 are the KAT frame, the others rnd(.., 1235 + i) against the oracle for two of them
    others = {3: orc.rnd(2160, 3840, 1238), 16: orc.rnd(2160, 3840, 1251)}
    batch = torch.empty((24, 2160, 3840, 3), dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    batch.copy_(torch.randint(0, 256, batch.shape, dtype=torch.uint8, device="cuda", generator=g))
    for i in (0, 7, 23):
        batch[i].copy_(x)
    for i, a in others.items():
        batch[i].copy_(torch.from_numpy(a))
    out = it.apply_dithering_frames(batch)
    for i in (0, 7, 23):
        assert H(out[i].cpu().numpy()) == case["h_out"], i
    for i, a in others.items():
        ref = orc.apply_dithering(a, pal, "error_diffusion", case["params"])
        assert np.array_equal(out[i].cpu().numpy(), ref), i


def test_c3_long_batch_of_4k_frames_on_persistent_workgroups(d, orc):
    """A batch larger than the device has CUs (300 4K frames, the product library): persistent workgroups, their waves running
    on from one frame into the next.  The reference's C3 frame sits at positions 0, 1, 255, 256 (first / second frame of a
    workgroup), 299 and must hash as the reference hashed it; a frame of another content (oracle) at 257."""
    import torch
    case = _need("c3_ed_fs_U16_rnd4k")
    arr = case_input(orc, case["input"])
    pal = case_palette(orc, case["palette"])
    it = d.ImageDitherer(len(pal), d.DitherMode.ERROR_DIFFUSION, pal, False, dict(case["params"]))
    n = 300
    batch = torch.empty((n, 2160, 3840, 3), dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    batch.copy_(torch.randint(0, 256, batch.shape, dtype=torch.uint8, device="cuda", generator=g))
    x = torch.from_numpy(arr).cuda()
    kat_at = (0, 1, 255, 256, 299)
    for i in kat_at:
        batch[i].copy_(x)
    other = orc.rnd(2160, 3840, 1260)
    batch[257].copy_(torch.from_numpy(other))
    out = it.apply_dithering_frames(batch)
    for i in kat_at:
        assert H(out[i].cpu().numpy()) == case["h_out"], i
    assert np.array_equal(out[257].cpu().numpy(), orc.apply_dithering(other, pal, "error_diffusion", case["params"]))
    # ... and every other frame equals what the same frame gives in a small batch (the few-frames schedule)
    for i in (2, 128, 258, 298):
        assert torch.equal(out[i], it.apply_dithering_frames(batch[i:i + 1])[0]), i


@pytest.mark.parametrize("K,variant", [(256, "floyd_steinberg"), (64, "atkinson")])
def test_error_diffusion_17_to_256_colours_4k_all_schedules(d, orc, K, variant):
    """The hierarchical nearest table at full size on the product library: one 4K frame (34 bands over 16 workgroups, table in LDS,
    8-step periods), and a batch of 280 frames (persistent sixteen-wave workgroups reading the table from L2) in which two contents
    alternate -- every frame of the batch against the oracle's result for its content (3 s per content on one core)."""
    import torch
    pal = orc.palr(K, 7)
    params = {"variant": variant, "serpentine": "false"}
    it = d.ImageDitherer(K, d.DitherMode.ERROR_DIFFUSION, pal, False, dict(params))
    a0, a1 = orc.rnd(2160, 3840, 4100 + K), orc.rnd(2160, 3840, 4200 + K)
    r0 = torch.from_numpy(orc.apply_dithering(a0, pal, "error_diffusion", params)).cuda()
    r1 = torch.from_numpy(orc.apply_dithering(a1, pal, "error_diffusion", params)).cuda()
    x0, x1 = torch.from_numpy(a0).cuda(), torch.from_numpy(a1).cuda()
    assert torch.equal(it.apply_dithering_frames(x0), r0)                       # one frame
    assert torch.equal(it.apply_dithering_frames(torch.stack([x0, x1, x0]))[1], r1)   # three frames: still the few-frames schedule
    n = 280
    batch = torch.empty((n, 2160, 3840, 3), dtype=torch.uint8, device="cuda")
    for i in range(n):
        batch[i].copy_(x1 if i % 3 == 1 else x0)
    out = it.apply_dithering_frames(batch)
    del batch
    for i in range(n):
        assert torch.equal(out[i], r1 if i % 3 == 1 else r0), i


@pytest.mark.parametrize("kind,K", [("smooth", 256), ("dark", 128)])
def test_error_diffusion_with_the_content_s_own_palette_4k(d, orc, kind, K):
    """A palette extracted from the content (median cut of the frame itself: crowded cells; the smooth one keeps the hierarchical table
    in LDS for few frames and the lists for batches, the dark one the lists and their octree everywhere): one 4K frame and a batch of 40
    against the oracle."""
    import torch
    from PIL import Image
    a0 = orc.imgl(2160, 3840, 71, kind)
    pal = d.ColorReducer.reduce_colors(Image.fromarray(a0, "RGB"), K)
    params = {"variant": "floyd_steinberg", "serpentine": "false"}
    it = d.ImageDitherer(K, d.DitherMode.ERROR_DIFFUSION, pal, False, dict(params))
    r0 = torch.from_numpy(orc.apply_dithering(a0, pal, "error_diffusion", params)).cuda()
    x0 = torch.from_numpy(a0).cuda()
    assert torch.equal(it.apply_dithering_frames(x0), r0)
    out = it.apply_dithering_frames(x0.unsqueeze(0).repeat(40, 1, 1, 1))
    for i in range(40):
        assert torch.equal(out[i], r0), i


@pytest.mark.parametrize("mode,params,K,gamma,kind", [
    ("perceptual", {}, 64, False, "rnd"), ("hybrid", {"lum_factor": 0.9, "col_factor": 0.4}, 256, False, "rnd"),
    ("adaptive_variance", {"var_threshold": 150.0, "window_radius": 2}, 100, False, "rnd"),
    ("ostromoukhov", {"serpentine": "false"}, 256, False, "rnd"),
    # the extended 16^3 lists (outermost cells unbounded): use_gamma crowds the palette at the dark faces of the cube, and a frame of
    # pure 0 / 255 values sends the accumulated errors beyond the cube at most pixels
    ("perceptual", {}, 128, True, "rnd"), ("hybrid", {"lum_factor": 1.3, "col_factor": 0.2}, 64, False, "extreme"),
    ("adaptive_variance", {"var_threshold": 80.0, "window_radius": 1}, 256, True, "extreme"), ("perceptual", {}, 17, False, "extreme")])
def test_variable_diffusers_1080p_more_than_16_colours(d, orc, mode, params, K, gamma, kind):
    """The variable-coefficient diffusers at 1080p (17 bands) with palettes whose candidate lists they read from L2 -- for the
    unclamped ones the extended 16^3 lists: one frame and a batch of 20 against the oracle, product library."""
    import torch
    pal = orc.palr(K, 5)
    a0 = orc.rnd(1080, 1920, 900 + K)
    if kind == "extreme":
        a0 = np.where(a0 < 128, 0, 255).astype(np.uint8)
    ref = torch.from_numpy(orc.apply_dithering(a0, pal, mode, params, gamma)).cuda()
    it = d.ImageDitherer(K, d.DitherMode(mode), pal, gamma, dict(params))
    x0 = torch.from_numpy(a0).cuda()
    assert torch.equal(it.apply_dithering_frames(x0), ref)
    out = it.apply_dithering_frames(x0.unsqueeze(0).repeat(20, 1, 1, 1))
    for i in range(20):
        assert torch.equal(out[i], ref), i


@pytest.mark.parametrize("K", [300, 1024])
def test_palettes_above_256_colours_1080p(d, orc, K):
    """257..1024 colours at 1080p on the product library with the accelerator built (cell table whose deeper split nodes stay in
    global memory) and without it (brute-force kernels), nearest / Bayer / IGN, and Floyd-Steinberg error diffusion, against the
    oracle."""
    import torch
    pal = orc.palr(K, 11)
    a0 = orc.rnd(1080, 1920, 300 + K)
    x0 = torch.from_numpy(a0).cuda()
    for mode, params in (("none", {}), ("bayer", {"size": "8x8"}), ("IGN", {"scale": 1.3, "seed": 5})):
        ref = torch.from_numpy(orc.apply_dithering(a0, pal, mode, params, False)).cuda()
        plain = d.ImageDitherer(K, d.DitherMode(mode), pal, False, dict(params))
        assert torch.equal(plain.apply_dithering_frames(x0), ref), (mode, "brute force")
        d.drop_device_caches()
        fast = d.ImageDitherer(K, d.DitherMode(mode), pal, False, dict(params)).prepare()
        assert torch.equal(fast.apply_dithering_frames(x0), ref), (mode, "accelerator")
        assert torch.equal(fast.apply_dithering_frames(x0.unsqueeze(0).repeat(5, 1, 1, 1))[4], ref), (mode, "batch")
    params = {"variant": "floyd_steinberg", "serpentine": "false"}
    ref = torch.from_numpy(orc.apply_dithering(a0, pal, "error_diffusion", params, False)).cuda()
    it = d.ImageDitherer(K, d.DitherMode.ERROR_DIFFUSION, pal, False, dict(params))
    assert torch.equal(it.apply_dithering_frames(x0), ref)
    assert torch.equal(it.apply_dithering_frames(x0.unsqueeze(0).repeat(3, 1, 1, 1))[2], ref)


def test_error_diffusion_gives_up_and_repairs(d, be, orc, switches):
    """A workgroup of a multi-workgroup launch that gives up waiting sets a flag; the repair launch behind it redoes the
    frame.  DP_ED_TEST_GIVEUP makes every workgroup of the first launch give up before it writes anything."""
    import torch
    pal = orc.generate_uniform_palette(16)
    frames = np.stack([orc.rnd(300, 200, 40 + i) for i in range(3)])  # 5 bands each: spread over workgroups
    x = torch.from_numpy(frames).cuda()
    switches.setenv("DP_ED_TEST_GIVEUP", "1")
    for mode, params in [("error_diffusion", {"variant": "floyd_steinberg", "serpentine": "false"}),
                         ("error_diffusion", {"variant": "jjn", "serpentine": "false"}),
                         ("perceptual", {}), ("hybrid", {}), ("ostromoukhov", {"serpentine": "false"})]:
        it = d.ImageDitherer(16, d.DitherMode(mode), pal, False, params)
        out = torch.full_like(x, 0x55)
        it.apply_dithering_frames(x, out=out)
        got = out.cpu().numpy()
        for i in range(3):
            assert np.array_equal(got[i], orc.apply_dithering(frames[i], pal, mode, params)), (mode, i)


# ------------------------------------------------------------------------------------------------ C4
def test_c4_8k_in_eight_row_bands(d, be, orc):
    """7680x4320 rnd(.., 99), blue noise (64, 42), 32 colours: dithered as 8 bands of 540 rows with global coordinates,
    the concatenation has the hash the reference computed for the whole image; Bayer and IGN bands against the oracle."""
    import torch
    case = _need("c4_blue64_p32_rnd8k")
    arr = case_input(orc, case["input"])
    assert H(arr) == case["h_in"]
    pal = case_palette(orc, case["palette"])
    from dither_pie_amd import sharding
    it = d.ImageDitherer(len(pal), d.DitherMode.BLUE_NOISE, pal, False, dict(case["params"]))
    outs = []
    for lo, hi in sharding.row_bands(4320, 8):
        outs.append(sharding.dither_band(it, torch.from_numpy(arr[lo:hi]).cuda(), lo).cpu().numpy())
    assert H(np.concatenate(outs)) == case["h_out"]
    for mode, params in [("bayer", {"size": "8x8"}), ("IGN", {"scale": 1.0, "seed": 3})]:
        it = d.ImageDitherer(len(pal), d.DitherMode(mode), pal, False, params)
        ref = orc.apply_dithering(arr, pal, mode, params)
        for lo, hi in sharding.row_bands(4320, 8)[2:5]:
            got = sharding.dither_band(it, torch.from_numpy(arr[lo:hi]).cuda(), lo).cpu().numpy()
            assert np.array_equal(got, ref[lo:hi]), (mode, lo)


def test_c4_kmeans_totals_over_all_8k_pixels(be, orc):
    """One Lloyd pass over all 33 M pixels of the C4 image: exact integer totals equal the oracle's, whole image and
    summed over 8 bands; then the full fit under a 1-rank NCCL (RCCL) group with the real kernel."""
    import torch
    arr = orc.rnd(4320, 7680, 99)
    px = torch.from_numpy(arr).cuda().reshape(-1, 3)
    rs = np.random.RandomState(1)
    centers = rs.rand(32, 3) * 255.0
    s_ref, n_ref, _ = orc.kmeans_step(arr.reshape(-1, 3), centers)
    s, n, q = be.kmeans_step(px, torch.from_numpy(centers))
    assert np.array_equal(s.cpu().numpy(), s_ref) and np.array_equal(n.cpu().numpy(), n_ref)
    x64 = arr.reshape(-1, 3).astype(np.int64)
    assert int(q.sum().item()) == int((x64 * x64).sum())
    from dither_pie_amd import sharding
    acc_s, acc_n = torch.zeros_like(s), torch.zeros_like(n)
    for lo, hi in sharding.row_bands(4320, 8):
        bs, bn, _ = be.kmeans_step(px[lo * 7680:hi * 7680], torch.from_numpy(centers))
        acc_s += bs
        acc_n += bn
    assert torch.equal(acc_s, s) and torch.equal(acc_n, n)
    # the same pass over the colour histogram of the 33 M pixels (built once; in 8 bands accumulated into one histogram too)
    hist = be.ColourHistogram(px)
    hs, hn, hq = hist.step(torch.from_numpy(centers))
    assert torch.equal(hs, s) and torch.equal(hn, n) and torch.equal(hq, q)
    hist8 = be.ColourHistogram(device=px.device)
    for i, (lo, hi) in enumerate(sharding.row_bands(4320, 8)):
        hist8.add(px[lo * 7680:hi * 7680], accumulate=i > 0)
    used = (1 << 26) + 4 * (4097 + 4096)   # the table, the cell totals, the occupied cells (all 4096 on noise)
    assert hist8.n == px.shape[0] and torch.equal(hist8.buf[:used], hist.buf[:used])
    mean = torch.from_numpy(orc.data_mean(arr.reshape(-1, 3)))
    ms_ref, mn_ref, _ = orc.kmeans_step(arr.reshape(-1, 3), np.round(centers), orc.data_mean(arr.reshape(-1, 3)))
    hs, hn, _ = hist8.step(torch.from_numpy(np.round(centers)), mean)
    assert np.array_equal(hs.cpu().numpy(), ms_ref) and np.array_equal(hn.cpu().numpy(), mn_ref)


def _hist_table_numpy(px):
    """count[colour] in the histogram's cell-major order, from numpy"""
    r, g, b = (px[:, i].astype(np.int64) for i in range(3))
    idx = ((r >> 4) << 20) | ((g >> 4) << 16) | ((b >> 4) << 12) | ((r & 15) << 8) | ((g & 15) << 4) | (b & 15)
    return np.bincount(idx, minlength=1 << 24).astype(np.uint32)


@pytest.mark.parametrize("kind", ["noise", "image", "flat", "few", "runs"])
def test_colour_histogram_counts(be, orc, kind):
    """dp_kmeans_hist_build_u8: count[colour] over all 2^24 colours equals numpy's, and the per-cell totals behind it, for
    noise (every pixel a new colour: the LDS merge table overflows into direct atomics), image-like content, one flat colour
    (every pixel the same address), a handful of colours, and runs; ragged pixel counts, an unaligned buffer, accumulation."""
    import torch
    rs = np.random.RandomState(5)
    n = 1_000_003
    if kind == "noise":
        px = rs.randint(0, 256, (n, 3)).astype(np.uint8)
    elif kind == "image":
        px = orc.imgl(700, 1000, 3, "smooth").reshape(-1, 3)
    elif kind == "flat":
        px = np.tile(np.array([[17, 99, 200]], np.uint8), (n, 1))
    elif kind == "few":
        px = rs.randint(0, 256, (6, 3)).astype(np.uint8)[rs.randint(0, 6, n)]
    else:
        px = np.repeat(rs.randint(0, 256, (n // 37 + 1, 3)).astype(np.uint8), 37, axis=0)[:n]
    t = torch.from_numpy(np.ascontiguousarray(px)).cuda()
    hist = be.ColourHistogram(t)
    table = hist.buf[:1 << 26].view(torch.int32).cpu().numpy().view(np.uint32)
    ref = _hist_table_numpy(px)
    assert np.array_equal(table, ref)
    info = hist.buf[1 << 26:].view(torch.int32).cpu().numpy().view(np.uint32)   # pixels per cell | occupied cells: how many, which
    per_cell = ref.reshape(4096, 4096).sum(1).astype(np.uint32)
    assert np.array_equal(info[:4096], per_cell)
    occupied = np.nonzero(per_cell)[0]
    assert info[4096] == len(occupied) and np.array_equal(info[4097:4097 + len(occupied)] & 0xfff, occupied)
    assert not (info[4097:4097 + len(occupied)] >> 31).any()   # (bit 31: a cell of 2^24 pixels and more)
    # an unaligned view (byte offset 3), a ragged count, accumulated on top
    raw = torch.empty(3 * px.shape[0] + 3, dtype=torch.uint8, device="cuda")
    raw[3:] = t.reshape(-1)
    part = raw[3:3 + 3 * 1001].view(-1, 3)
    assert part.data_ptr() % 4 != 0
    hist.add(part, accumulate=True)
    table2 = hist.buf[:1 << 26].view(torch.int32).cpu().numpy().view(np.uint32)
    assert np.array_equal(table2, ref + _hist_table_numpy(px[:1001]))
    # nothing at all
    empty = be.ColourHistogram(t[:0])
    assert int(empty.buf[:(1 << 26) + 4 * 4097].view(torch.int32).abs().sum().item()) == 0


def test_fused_lloyd_iteration_soak_in_the_suite():
    """The one-launch Lloyd iteration (hist_pass_kernel<FUSE>: totals by atomics, the last ticket runs the centre update, no
    fence -- outside the HIP memory model, gfx950 only) against the three-step iteration and the pass over the pixels, fit after
    fit: an ordering bug between the totals and the ticket shows as a fit that differs.  A short soak of tests/fuzz_kmeans_fused.py
    on every run of the suite (round-4 advisor), so that a toolchain or driver change is caught; the long one runs by hand."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_kmeans_fused.py")
    spec = importlib.util.spec_from_file_location("fuzz_kmeans_fused", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(5) == 0


def test_colour_histogram_flags_a_wrapped_cell_count(be):
    """dp_kmeans_hist_build_u8 with accumulate = 1 checks the running 32-bit cell totals on the device (round-4 advisor): a cell
    whose count passes 2^32 sets the overflow word of the info block, which stays set until a build that does not accumulate.
    (The Python wrapper's own guard counts pixels on the host; here the cell total is pushed to the brink directly.)"""
    import torch
    px = torch.tensor([[17, 99, 200]] * 40, dtype=torch.uint8, device="cuda")
    hist = be.ColourHistogram(px)
    assert not hist.overflowed()
    cell = (17 >> 4) << 8 | (99 >> 4) << 4 | (200 >> 4)
    info = hist.buf[1 << 26:].view(torch.int32)
    assert int(info[cell].item()) == 40
    info[cell] = -20                      # 2^32 - 20 pixels in that cell already
    hist.add(px[:10], accumulate=True)    # 2^32 - 10: fine
    assert not hist.overflowed()
    hist.add(px[:30], accumulate=True)    # wraps
    assert hist.overflowed()
    hist.add(px[:5], accumulate=True)     # sticky
    assert hist.overflowed()
    hist.add(px, accumulate=False)
    assert not hist.overflowed() and int(info[cell].item()) == 40


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_kmeans_histogram_pass_fuzz(be, orc, seed):
    """Random cluster counts (1..256), centre layouts (float, integer = tie-rich, data points, crowded), pixel counts and
    content through the histogram pass: the int64 totals equal the oracle's float64 labelling every time, with and without
    sklearn's rule for equidistant colours -- the generator of test_kmeans_cell_list_fuzz, on the product library."""
    import torch
    rs = np.random.RandomState(3000 + seed)
    for case in range(12):
        K = int(rs.choice([1, 2, 3, 7, 16, 31, 32, 64, 65, 129, 255, 256]))
        n = int(rs.choice([1, 3, 255, 257, 4096, 50001, 200003]))
        kind = rs.randint(0, 4)
        if kind == 0:
            px = rs.randint(0, 256, (n, 3)).astype(np.uint8)
        elif kind == 1:
            t = np.arange(n)
            px = np.clip(np.stack([t * 255 // max(n - 1, 1), 255 - t * 255 // max(n - 1, 1), (t // 7) % 256], -1) + rs.randint(-2, 3, (n, 3)), 0, 255).astype(np.uint8)
        elif kind == 2:
            px = rs.randint(0, 256, (5, 3)).astype(np.uint8)[rs.randint(0, 5, n)]
        else:
            px = rs.randint(0, 40, (n, 3)).astype(np.uint8)
        ckind = rs.randint(0, 4)
        if ckind == 0:
            centers = rs.rand(K, 3) * 255.0
        elif ckind == 1:
            centers = np.round(rs.rand(K, 3) * 255.0)
        elif ckind == 2:
            centers = px[rs.randint(0, n, K)].astype(np.float64) + rs.choice([0.0, 0.5, 0.25])
        else:
            centers = 5.0 + rs.rand(K, 3) * 20.0
        mean = orc.data_mean(px) if rs.rand() < 0.5 else None
        s_ref, n_ref, _ = orc.kmeans_step(px, centers, mean)
        hist = be.ColourHistogram(torch.from_numpy(px).cuda())
        s, cnt, q = hist.step(torch.from_numpy(centers), None if mean is None else torch.from_numpy(mean))
        assert np.array_equal(s.cpu().numpy(), s_ref) and np.array_equal(cnt.cpu().numpy(), n_ref), (seed, case, K, n, kind, ckind)
        x64 = px.astype(np.int64)
        assert int(q.sum().item()) == int((x64 * x64).sum())


def test_kmeans_histogram_pass_giant_cell(be, orc):
    """A cell of the colour cube with 2^24 pixels and more (a flat 8K frame, a letterboxed video): its weighted sums leave 32
    bits, the pass takes 64-bit partials for it (bit 31 of the occupied-cell entry) -- totals against the oracle, one and several
    candidates in that cell, next to ordinary cells."""
    import torch
    n_big = (1 << 24) + (1 << 20) + 12345
    rs = np.random.RandomState(4)
    big = np.empty((n_big, 3), np.uint8)
    big[:] = (250, 245, 240)
    big[::3] = (253, 246, 245)                     # the same 16^3 cell, another colour
    big[1::7] = (255, 255, 253)                    # ... and a third, near the cell's far corner
    rest = rs.randint(0, 256, (70001, 3)).astype(np.uint8)
    px = np.concatenate([big, rest])
    hist = be.ColourHistogram(torch.from_numpy(px).cuda())
    info = hist.buf[1 << 26:].view(torch.int32).cpu().numpy().view(np.uint32)
    cell = (250 >> 4) << 8 | (245 >> 4) << 4 | (240 >> 4)
    assert (info[4097:4097 + info[4096]] >> 31).sum() == 1 and info[cell] >= (1 << 24)
    x64 = px.astype(np.int64)
    for centers in (np.array([[128.0, 128.0, 128.0]]),                                     # one candidate everywhere
                    np.array([[251.0, 245.0, 241.0], [254.0, 253.0, 251.0], [40.0, 40.0, 40.0], [250.0, 20.0, 220.0]])):  # two inside the giant cell
        s_ref, n_ref, q_ref = orc.kmeans_step(px, centers)
        s, cnt, q = hist.step(torch.from_numpy(centers))
        assert np.array_equal(s.cpu().numpy(), s_ref) and np.array_equal(cnt.cpu().numpy(), n_ref)
        assert int(q.sum().item()) == int((x64 * x64).sum())
        assert int(n_ref.max()) >= (1 << 23)
    assert int(s_ref.sum()) > (1 << 32)   # (sums beyond 32 bits did occur)


def test_lloyd_over_the_histogram_equals_lloyd_over_the_pixels(be, orc, gold, kat):
    """kmeans.lloyd with histogram=True against histogram=False on the reference's k-means fixtures (same seeds, sklearn's tie
    rule): identical centres, inertia and iteration counts -- and therefore the reference's palettes; and on 2^20 pixels of
    image-like content, where the histogram is what lloyd() takes by itself."""
    import torch
    from dither_pie_amd import kmeans
    for nm, m in sorted(kat["misc"]["kmeans_extra"].items()):
        arr = case_input(orc, m["input"])
        px = torch.from_numpy(arr).cuda().reshape(-1, 3)
        init = arr.reshape(-1, 3)[gold[f"{nm}_init_idx"]].astype(np.float64)
        a = kmeans.lloyd(px, init, histogram=True)                  # one launch per iteration (dp_kmeans_hist_iterate)
        b = kmeans.lloyd(px, init, histogram=False)
        c = kmeans.lloyd(px, init, histogram=True, fuse=False)      # pass / update as separate launches (a sharded fit's steps)
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2] == m["n_iter"], nm
        assert np.array_equal(a[0], c[0]) and a[1] == c[1] and a[2] == c[2], nm
        assert np.abs(a[0] - gold[f"{nm}_centers"]).max() < 1e-9, nm
    big = orc.imgl(1024, 1024, 9, "smooth")
    px = torch.from_numpy(big).cuda().reshape(-1, 3)
    init = orc.kmeans_plusplus(big.reshape(-1, 3)[::97], 24, np.random.RandomState(3))
    assert px.shape[0] >= kmeans.HIST_MIN_PIXELS
    a = kmeans.lloyd(px, init)                    # (the histogram by default)
    b = kmeans.lloyd(px, init, histogram=False)
    assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2]


@pytest.mark.parametrize("K", [1, 5, 32, 33, 100, 256])
def test_kmeans_matrix_core_kernel_totals(be, orc, switches, K):
    """The opt-in matrix-core Lloyd pass (kmeans_mfma_kernel, DP_KMEANS_MFMA=1; measured slower, kept as evidence):
    same integer totals as the oracle and as the product kernel, ragged pixel count, centres on and off the lattice."""
    import torch
    arr = orc.rnd(517, 1031, 7 + K)
    flat = arr.reshape(-1, 3)[:517 * 1031 - 3]
    px = torch.from_numpy(np.ascontiguousarray(flat)).cuda()
    rs = np.random.RandomState(K)
    centers = rs.rand(K, 3) * 255.0
    centers[: K // 2] = np.round(centers[: K // 2])  # integer centres: exact ties between clusters do occur
    s_ref, n_ref, _ = orc.kmeans_step(flat, centers)
    switches.setenv("DP_KMEANS_MFMA", "1")
    s, n, q = be.kmeans_step(px, torch.from_numpy(centers))
    switches.delenv("DP_KMEANS_MFMA")
    s2, n2, q2 = be.kmeans_step(px, torch.from_numpy(centers))
    assert np.array_equal(s.cpu().numpy(), s_ref) and np.array_equal(n.cpu().numpy(), n_ref)
    assert torch.equal(s, s2) and torch.equal(n, n2) and torch.equal(q, q2)


@pytest.mark.parametrize("K,kind", [(1, "random"), (2, "lattice"), (5, "random"), (16, "clustered"), (32, "lattice"), (33, "data"),
                                    (64, "random"), (65, "lattice"), (100, "clustered"), (200, "data"), (255, "random"), (256, "data"),
                                    (256, "clustered"), (8, "duplicates"), (130, "duplicates")])
def test_kmeans_cell_list_kernel_totals(be, orc, switches, K, kind):
    """The Lloyd pass over per-cell candidate lists (kmeans_cells_kernel, what images above 2^19 pixels take; forced here
    with DP_KMEANS_CELLS=1): integer totals equal the oracle's and the full-scan kernel's -- ragged pixel count, an
    unaligned pixel pointer, centres on the integer lattice (exact ties between clusters), centres crowded into one
    corner of the cube (cells with more candidates than a list holds), duplicated centres, smooth content (whole waves
    with one label: the wave-level sums) and uniform noise (the packed LDS atomics)."""
    import torch
    rs = np.random.RandomState(100 + K)
    noise = orc.rnd(411, 1031, 17 + K).reshape(-1, 3)
    yy, xx = np.mgrid[0:300, 0:1031]
    smooth = np.stack([xx * 255 // 1030, yy * 255 // 299, (xx + yy) * 255 // 1329], -1).reshape(-1, 3)
    smooth = np.clip(smooth + rs.randint(-3, 4, smooth.shape), 0, 255).astype(np.uint8)
    flat = np.concatenate([noise, smooth])[: len(noise) + len(smooth) - 3]
    if kind == "random":
        centers = rs.rand(K, 3) * 255.0
    elif kind == "lattice":
        centers = np.round(rs.rand(K, 3) * 255.0)
    elif kind == "clustered":
        centers = 20.0 + rs.rand(K, 3) * 30.0
    elif kind == "data":
        centers = flat[rs.randint(0, len(flat), K)].astype(np.float64) + 0.25
    else:
        centers = np.repeat(np.round(rs.rand(K // 2, 3) * 255.0), 2, axis=0)
    s_ref, n_ref, _ = orc.kmeans_step(flat, centers)
    buf = torch.zeros(len(flat) * 3 + 1, dtype=torch.uint8, device="cuda")
    for offset in (0, 1):  # 4-byte aligned, then not
        px = buf[offset:offset + len(flat) * 3].view(-1, 3)
        px.copy_(torch.from_numpy(np.ascontiguousarray(flat)))
        switches.setenv("DP_KMEANS_CELLS", "1")
        s, n, q = be.kmeans_step(px, torch.from_numpy(centers))
        switches.setenv("DP_KMEANS_CELLS", "0")
        s2, n2, q2 = be.kmeans_step(px, torch.from_numpy(centers))
        switches.delenv("DP_KMEANS_CELLS")
        assert np.array_equal(s.cpu().numpy(), s_ref) and np.array_equal(n.cpu().numpy(), n_ref), (K, kind, offset)
        assert torch.equal(s, s2) and torch.equal(n, n2) and torch.equal(q, q2), (K, kind, offset)
        tot = torch.zeros(5 * K, dtype=torch.int64, device="cuda")
        switches.setenv("DP_KMEANS_CELLS", "1")
        be.kmeans_step_into(px, torch.from_numpy(centers).cuda(), tot, want_sq=False)  # the instance without squared norms
        switches.delenv("DP_KMEANS_CELLS")
        assert np.array_equal(tot[:3 * K].cpu().numpy().reshape(K, 3), s_ref) and np.array_equal(tot[3 * K:4 * K].cpu().numpy(), n_ref)


@pytest.mark.parametrize("n,K,seed", [(10000, 32, 42), (10000, 256, 1), (9999, 16, 7), (517, 8, 3), (64, 64, 5), (40, 5, 11), (16384, 2, 2), (3, 1, 0)])
def test_kmeans_plusplus_kernel_equals_host_seeding(be, n, K, seed):
    """dp_kmeans_plusplus_u8 (one workgroup, integer arithmetic) against the product's numpy statement of sklearn's
    _kmeans_plusplus (kmeans.kmeans_plusplus) on random samples, on samples full of duplicates (zero distances, equal
    potentials) and at the LDS capacity.  This is kernel == host statement; that the statement (and the kernel) pick what
    sklearn picks is pinned separately, against sklearn's own indices: test_oracle_golden.py::
    test_kmeans_plusplus_picks_sklearns_seeds (host) and test_kmeans_plusplus_kernel_picks_sklearns_seeds below."""
    import torch
    from dither_pie_amd import kmeans
    rs = np.random.RandomState(seed)
    sample = rs.randint(0, 256, (n, 3)).astype(np.uint8)
    if n in (64, 40):
        sample = sample[rs.randint(0, max(K, 6), n)]  # few distinct colours: candidates at distance 0, tied potentials
    host = kmeans.kmeans_plusplus(sample, K, np.random.RandomState(seed))
    dev = kmeans.kmeans_plusplus_device(torch.from_numpy(sample).cuda(), K, np.random.RandomState(seed))
    assert np.array_equal(dev.cpu().numpy(), host)


def _km_fixture_cases(kat):
    out = []
    for nm in ("km8", "km16", "km32"):
        m = kat["misc"][nm]
        out.append((nm, ["rnd", m["h"], m["w"], m["seed"]] if m["kind"] == "rnd" else ["grad", m["h"], m["w"]], m["K"], 42, m))
    for nm, m in sorted(kat["misc"]["kmeans_extra"].items()):
        out.append((nm, m["input"], m["K"], m["random_state"], m))
    return out


def test_kmeans_plusplus_kernel_picks_sklearns_seeds(be, orc, gold, kat):
    """The seeding kernel picks exactly the sample indices sklearn's _kmeans_plusplus picked (km*_init_idx, kmx_*_init_idx:
    produced by sklearn itself in make_golden.py) on all 11 reference k-means inputs - first centre drawn as
    RandomState.choice(n, p=uniform), the trials' uniforms in sklearn's order."""
    import torch
    from conftest import case_input
    from dither_pie_amd import kmeans
    for nm, spec, K, rs_seed, _ in _km_fixture_cases(kat):
        px = case_input(orc, spec).reshape(-1, 3)
        n = len(px)
        rs = np.random.RandomState(rs_seed)
        n_trials = 2 + int(np.log(K))
        first = kmeans.first_center_draw(n, rs)
        uniforms = np.stack([rs.uniform(size=n_trials) for _ in range(1, K)]) if K > 1 else np.zeros((0, n_trials))
        ids, centers = be.kmeans_plusplus(torch.from_numpy(px).cuda(), K, first, uniforms)
        assert np.array_equal(ids.cpu().numpy(), gold[f"{nm}_init_idx"]), nm
        assert np.array_equal(centers.cpu().numpy(), px[gold[f"{nm}_init_idx"]].astype(np.float64)), nm


@pytest.mark.parametrize("variant,K", [("scan", 16), ("scan", 300), ("cells", 16), ("cells", 100), ("mfma", 40)])
def test_kmeans_pass_sklearn_tie_rule(be, orc, switches, variant, K):
    """dp_kmeans_step_u8 with mean_dev: pixels equidistant from two centres get the label of sklearn's float64
    expression on mean-centred data (the oracle's orc_kmeans_step_sk, pinned by the kmx_* fixtures), in every kernel
    that has a float64 decision (full scan with and without keys, both candidate-list widths, the matrix-core kernel).
    Centres are data points of a structured image, as in the first pass after k-means++: ~1 % exact ties."""
    import torch
    yy, xx = np.mgrid[0:257, 0:403]
    img = np.stack([xx % 256, yy % 256, ((xx + yy) // 2) % 256], -1).astype(np.uint8)
    flat = np.ascontiguousarray(np.concatenate([img.reshape(-1, 3), orc.rnd(64, 403, K).reshape(-1, 3) // 8 * 8])[:-1])
    rs = np.random.RandomState(K)
    centers = flat[rs.randint(0, len(flat), K)].astype(np.float64)
    mean = orc.data_mean(flat)
    s_ref, n_ref, _ = orc.kmeans_step(flat, centers, mean)
    s_low, n_low, _ = orc.kmeans_step(flat, centers)
    assert not np.array_equal(n_ref, n_low), "the input has to tell the two tie rules apart"
    px = torch.from_numpy(flat).cuda()
    if variant == "cells":
        switches.setenv("DP_KMEANS_CELLS", "1")
    elif variant == "mfma":
        switches.setenv("DP_KMEANS_MFMA", "1")
    else:
        switches.setenv("DP_KMEANS_CELLS", "0")
    s, n, q = be.kmeans_step(px, torch.from_numpy(centers), torch.from_numpy(mean))
    s0, n0, _ = be.kmeans_step(px, torch.from_numpy(centers))
    assert np.array_equal(s.cpu().numpy(), s_ref) and np.array_equal(n.cpu().numpy(), n_ref)
    assert np.array_equal(s0.cpu().numpy(), s_low) and np.array_equal(n0.cpu().numpy(), n_low)
    assert int(q.sum()) == int((flat.astype(np.int64) ** 2).sum())


def test_fit_palette_under_one_rank_nccl_group(be, orc):
    """kmeans.fit_palette with torch.distributed initialised on the RCCL backend (world size 1): the all-reduce of the
    integer totals runs through RCCL on device tensors produced by the real kernel, and changes nothing."""
    import socket
    import torch
    import torch.distributed as dist
    from dither_pie_amd import kmeans
    arr = orc.rnd(300, 400, 21)
    px = torch.from_numpy(arr).cuda().reshape(-1, 3)
    pal0, c0, i0, n0 = kmeans.fit_palette(px, 16, 42)
    if dist.is_initialized():
        pytest.skip("a process group is already active")
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        pal1, c1, i1, n1 = kmeans.fit_palette(px, 16, 42, group=dist.group.WORLD)
    finally:
        dist.destroy_process_group()
    assert pal1 == pal0 and np.array_equal(c1, c0) and i1 == i0 and n1 == n0
    init = kmeans.kmeans_plusplus_device(kmeans.seed_sample(px, px.shape[0], 0, 42, as_tensor=True), 16, np.random.RandomState(42))
    c_or, i_or, _ = orc.kmeans_lloyd(arr.reshape(-1, 3), init.cpu().numpy())
    assert np.abs(c_or - c1).max() < 1e-9
    # the device seeding is the host seeding (same draws, same arithmetic up to the order of the prefix sum)
    init_h = kmeans.kmeans_plusplus(kmeans.seed_sample(px, px.shape[0], 0, 42), 16, np.random.RandomState(42))
    assert np.abs(init.cpu().numpy() - init_h).max() == 0.0


def _two_rank_fit_worker(rank, world, port, h, w, K, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dither_pie_amd import kmeans, sharding
        from oracle import oracle as orc
        torch.cuda.set_device(0)
        img = orc.rnd(h, w, 77)
        lo, hi = sharding.shard_range(h, rank, world)
        band = torch.from_numpy(np.ascontiguousarray(img[lo:hi])).cuda()
        pal, centers, inertia, n_iter = kmeans.fit_palette(band.reshape(-1, 3), K, 42, n_total=h * w, offset=lo * w)
        # C4's second half and C5's partition with the real kernels: the band with global coordinates, a block of frames
        from dither_pie_amd.dithering_lib import DitherMode, ImageDitherer
        d4 = ImageDitherer(K, DitherMode.BLUE_NOISE, pal, False, {"size": 32, "seed": 7})
        whole = sharding.gather_bands(sharding.dither_band(d4, band, lo), h)
        frames = np.stack([orc.rnd(40, 56, 500 + i) for i in range(5)])
        d5 = ImageDitherer(16, DitherMode.BAYER, orc.generate_uniform_palette(16), False, {"size": "4x4"})
        flo, fhi = sharding.my_frame_block(5)
        video = sharding.gather_frames(sharding.dither_frames_sharded(d5, torch.from_numpy(frames[flo:fhi]).cuda()), 5)
        q.put((rank, pal, np.asarray(centers), float(inertia), int(n_iter), whole.cpu().numpy(), video.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_fit_palette_two_ranks_real_kernel_equals_one_rank(be, orc):
    """The sharded paths with the REAL kernels on two ranks (two processes sharing this GPU; the collectives run through a
    gloo group on device tensors -- RCCL itself needs one GPU per rank): C4 = seeding sample gathered from both bands,
    k-means++ kernel on each rank, Lloyd passes + all-reduce + device-side update, then the blue-noise dither of each band
    with global coordinates, gathered; C5 = contiguous frame blocks, gathered.  Byte-identical to the single-rank results."""
    import socket
    import torch
    import torch.multiprocessing as mp
    from dither_pie_amd import kmeans
    h, w, K = 301, 333, 12
    img = orc.rnd(h, w, 77)
    pal1, c1, i1, n1 = kmeans.fit_palette(torch.from_numpy(img).cuda().reshape(-1, 3), K, 42)
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_fit_worker, args=(r, 2, port, h, w, K, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref_img = orc.apply_dithering(img, pal1, "blue_noise", {"size": 32, "seed": 7})
    frames = np.stack([orc.rnd(40, 56, 500 + i) for i in range(5)])
    ref_video = np.stack([orc.apply_dithering(f, orc.generate_uniform_palette(16), "bayer", {"size": "4x4"}) for f in frames])
    for rank, pal, c, i, n, whole, video in res:
        assert pal == pal1 and np.array_equal(c, np.asarray(c1)) and i == float(i1) and n == int(n1), rank
        assert np.array_equal(whole, ref_img) and np.array_equal(video, ref_video), rank


# ------------------------------------------------------------------------------------------------ C5 / C1 / C2
def test_c5_1080p_frames_in_a_batch(d, orc):
    import torch
    case = _need("c5_bayer4_U16_rnd1080")
    pal = case_palette(orc, case["palette"])
    it = d.ImageDitherer(len(pal), d.DitherMode.BAYER, pal, False, dict(case["params"]))
    frames = np.stack([orc.rnd(1080, 1920, i) for i in range(4)])
    out = it.apply_dithering_frames(torch.from_numpy(frames).cuda()).cpu().numpy()
    assert H(frames[0]) == case["h_in"] and H(out[0]) == case["h_out"]
    for i in (1, 3):
        assert np.array_equal(out[i], orc.apply_dithering(frames[i], pal, "bayer", case["params"])), i


def test_c1_image_basic_exact(d, orc):
    """examples/image_basic.json as the reference runs it: palette=None => median cut of the image to 16 colours, Bayer
    default 4x4; the palette the object keeps and the output equal the reference's."""
    from PIL import Image
    case = _need("c1_bayer4_mc16_rnd512")
    arr = case_input(orc, case["input"])
    it = d.ImageDitherer(16, d.DitherMode.BAYER, None, False, dict(case["params"]))
    out = np.array(it.apply_dithering(Image.fromarray(arr)))
    assert [list(map(int, c)) for c in it.palette] == case["palette"][1]
    assert H(out) == case["h_out"]


# ------------------------------------------------------------------------------------------------ robustness
def test_out_buffer_is_validated(d, be, orc):
    import torch
    P = be.Palette(*orc.prepare_palette(orc.palr(16), False))
    thr = be.Thresholds.from_matrix(orc.bayer_matrix("4x4"))
    x = torch.from_numpy(orc.rnd(20, 30, 1)).cuda()
    ref = be.ordered(x, P, be.MODE_MATRIX, thr=thr)
    assert ref.shape == x.shape
    out3 = torch.empty_like(x)
    got = be.ordered(x, P, be.MODE_MATRIX, thr=thr, out=out3)           # 3-D out for a 3-D frame: the image, not a row
    assert got.shape == x.shape and torch.equal(got, ref) and got.data_ptr() == out3.data_ptr()
    with pytest.raises(ValueError):
        be.ordered(x, P, be.MODE_MATRIX, thr=thr, out=torch.empty((20, 31, 3), dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError):
        be.ordered(x, P, be.MODE_MATRIX, thr=thr, out=torch.empty((40, 30, 3), dtype=torch.uint8, device="cuda")[::2])
    with pytest.raises(TypeError):
        be.ordered(x, P, be.MODE_MATRIX, thr=thr, out=torch.empty((20, 30, 3), dtype=torch.float32, device="cuda"))
    with pytest.raises(TypeError):
        be.ordered(x, P, be.MODE_MATRIX, thr=thr, out=torch.empty((20, 30, 3), dtype=torch.uint8))
    taps, div = orc.ed_kernel("floyd_steinberg")
    with pytest.raises(ValueError):
        be.error_diffusion(x, P, taps, div, False, out=torch.empty((10, 30, 3), dtype=torch.uint8, device="cuda"))


def test_two_threads_share_a_stream(d, orc):
    """The GUI calls apply_dithering from worker threads on the default stream: two threads hammering different modes,
    palettes and sizes get exactly the single-threaded results (workspace + launch sequences are serialised per stream,
    accelerator builds and the device-object caches are locked)."""
    import torch
    jobs = []
    for k, (mode, params, K, shape, seed) in enumerate([
            ("bayer", {"size": "8x8"}, 256, (700, 1600), 1), ("error_diffusion", {"variant": "floyd_steinberg"}, 16, (300, 260), 2),
            ("none", {}, 64, (512, 2048), 3), ("blue_noise", {"size": 32, "seed": 1}, 32, (900, 1300), 4),
            ("IGN", {}, 256, (640, 1800), 5), ("hybrid", {}, 16, (280, 300), 6)]):
        arr = orc.rnd(shape[0], shape[1], seed)
        pal = orc.palr(K, 20 + k)   # fresh palettes: both threads trigger accelerator builds
        jobs.append((mode, params, pal, arr, orc.apply_dithering(arr, pal, mode, params)))
    errors = []

    def worker(order):
        try:
            for rep in range(3):
                for j in order:
                    mode, params, pal, arr, ref = jobs[j]
                    it = d.ImageDitherer(len(pal), d.DitherMode(mode), pal, False, params)
                    out = it.apply_dithering_frames(torch.from_numpy(arr).cuda()).cpu().numpy()
                    if not np.array_equal(out, ref):
                        errors.append((mode, rep))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(o,)) for o in ([0, 1, 2, 3, 4, 5], [5, 4, 3, 2, 1, 0])]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_kmeans_cell_list_fuzz(be, orc, switches, seed):
    """Random cluster counts (1..256), centre layouts, pixel counts and content through the candidate-list pass: the
    int64 totals equal the oracle's float64 labelling every time."""
    import torch
    rs = np.random.RandomState(1000 + seed)
    switches.setenv("DP_KMEANS_CELLS", "1")
    for case in range(12):
        K = int(rs.choice([1, 2, 3, 7, 16, 31, 32, 64, 65, 129, 255, 256]))
        n = int(rs.choice([1, 3, 255, 257, 4096, 50001, 200003]))
        kind = rs.randint(0, 4)
        if kind == 0:      # noise
            px = rs.randint(0, 256, (n, 3)).astype(np.uint8)
        elif kind == 1:    # a ramp with grain: long runs of one label
            t = np.arange(n)
            px = np.clip(np.stack([t * 255 // max(n - 1, 1), 255 - t * 255 // max(n - 1, 1), (t // 7) % 256], -1) + rs.randint(-2, 3, (n, 3)), 0, 255).astype(np.uint8)
        elif kind == 2:    # a few distinct colours only
            px = rs.randint(0, 256, (5, 3)).astype(np.uint8)[rs.randint(0, 5, n)]
        else:              # dark content: everything in one corner of the cube
            px = rs.randint(0, 40, (n, 3)).astype(np.uint8)
        ckind = rs.randint(0, 4)
        if ckind == 0:
            centers = rs.rand(K, 3) * 255.0
        elif ckind == 1:
            centers = np.round(rs.rand(K, 3) * 255.0)
        elif ckind == 2:
            centers = px[rs.randint(0, n, K)].astype(np.float64) + rs.choice([0.0, 0.5, 0.25])
        else:
            centers = 5.0 + rs.rand(K, 3) * 20.0
        # half of the cases with sklearn's tie rule (mean_dev: labels of equidistant pixels from sklearn's float64 expression)
        mean = orc.data_mean(px) if rs.rand() < 0.5 else None
        s_ref, n_ref, _ = orc.kmeans_step(px, centers, mean)
        s, cnt, q = be.kmeans_step(torch.from_numpy(px).cuda(), torch.from_numpy(centers), None if mean is None else torch.from_numpy(mean))
        assert np.array_equal(s.cpu().numpy(), s_ref) and np.array_equal(cnt.cpu().numpy(), n_ref), (seed, case, K, n, kind, ckind)
        x64 = px.astype(np.int64)
        assert int(q.sum().item()) == int((x64 * x64).sum())


def test_two_threads_share_a_stream_for_lloyd_passes(be, orc, switches):
    """dp_kmeans_step_u8 with per-cell candidate lists keeps the lists in library-owned memory per (device, stream): two
    threads on the default stream with different centres (ctypes drops the GIL) must not see each other's lists."""
    import torch
    switches.setenv("DP_KMEANS_CELLS", "1")
    px_np = orc.rnd(600, 1000, 77).reshape(-1, 3)
    px = torch.from_numpy(px_np).cuda()
    sets = []
    for i, K in enumerate([32, 7, 100, 32]):
        c = np.random.RandomState(300 + i).rand(K, 3) * 255.0
        sets.append((torch.from_numpy(c).cuda(), orc.kmeans_step(px_np, c)))
    errors = []

    def worker(order):
        try:
            for rep in range(40):
                for j in order:
                    c, (s_ref, n_ref, _) = sets[j]
                    s, n, _q = be.kmeans_step(px, c)
                    if not (np.array_equal(s.cpu().numpy(), s_ref) and np.array_equal(n.cpu().numpy(), n_ref)):
                        errors.append((j, rep))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(o,)) for o in ([0, 1, 2, 3], [3, 2, 1, 0])]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:5]


def test_frames_over_devices_in_process(d, orc):
    """sharding.process_on_devices: contiguous frame blocks, one worker thread and stream per device entry (the one GPU
    of the test box named twice), result equal to the single-call result and in frame order."""
    import torch
    from dither_pie_amd import sharding, video_processor as v
    pal = orc.palr(16, 3)
    it = d.ImageDitherer(16, d.DitherMode.BAYER, pal, False, {"size": "4x4"})
    frames = torch.from_numpy(np.stack([orc.rnd(90, 120, i) for i in range(11)]))
    ref = v.process_frames(frames.cuda(), it, "regular", 32, 2).cpu()
    for devs in ([0], [0, 0], [0, 0, 0]):
        got = sharding.process_on_devices(frames, lambda x: v.process_frames(x, it, "regular", 32, 2), devs, chunk=3)
        assert got.shape == ref.shape and torch.equal(got, ref), devs


def test_c2_4k_with_and_without_accelerator(d, orc):
    """BASELINE config 2 through the drop-in API, both ways a palette can serve it: the first frames of a new palette run
    on the brute-force kernels + fix-up pass (one image never pays for the accelerator), prepare() / enough pixels switch
    to the LDS cell-table kernels; both give the reference's bytes.  Likewise nearest-only and IGN."""
    import torch
    case = _need("bayer8_p256_rnd4k")
    arr = case_input(orc, case["input"])
    pal = [tuple(c) for c in orc.palr(256, 7)]
    x = torch.from_numpy(arr).cuda()
    pal_b = list(pal)
    it = d.ImageDitherer(256, d.DitherMode.BAYER, pal_b, False, dict(case["params"]))
    from dither_pie_amd.dithering_lib import _device_palette, prepare_palette
    P = _device_palette(*prepare_palette(pal_b, False))
    if not P._accel_done:
        assert H(it.apply_dithering_frames(x).cpu().numpy()) == case["h_out"]      # brute force + fix-up
    it.prepare()
    assert P._accel_done
    assert H(it.apply_dithering_frames(x).cpu().numpy()) == case["h_out"]          # cell-table kernel
    for mode, params in [("none", {}), ("IGN", {"scale": 1.0, "seed": 3}), ("bayer", {"size": "4x4"})]:
        it2 = d.ImageDitherer(256, d.DitherMode(mode), pal_b, False, params)
        assert np.array_equal(it2.apply_dithering_frames(x[:700]).cpu().numpy(), orc.apply_dithering(arr[:700], pal, mode, params)), mode


def test_accelerator_is_built_once_the_pixels_paid_for_it(d, be, orc):
    import torch
    pal = orc.palr(256, 31)
    P = be.Palette(*orc.prepare_palette(pal, False))
    thr = be.Thresholds.from_matrix(orc.bayer_matrix("8x8"))
    assert 4e7 < P.accel_break_even_pixels() < 7e7 and not P._accel_done   # (3.5 ms of build against 0.063 ns per pixel)
    x = torch.from_numpy(orc.rnd(2160, 3840, 5)).cuda().unsqueeze(0).repeat(3, 1, 1, 1)   # 25 Mpixel per call
    ref = orc.apply_dithering(orc.rnd(2160, 3840, 5), pal, "bayer", {"size": "8x8"})
    for call in range(4):
        out = be.ordered(x, P, be.MODE_MATRIX, thr=thr)
        assert P._accel_done == (call >= 2), call      # 25, 50 Mpixel served by brute force; the third call builds it
        assert np.array_equal(out[call % 3].cpu().numpy(), ref)
    big = be.Palette(*orc.prepare_palette(orc.palr(1024, 3), False))
    assert 8e7 < big.accel_break_even_pixels() < 1.2e8    # (25 ms of build at 1024 colours)
