"""GPU tier: the drop-in classes (ImageDitherer, strategies, ColorReducer, VideoProcessor helpers) end to end
against the reference-generated fixtures and the oracle."""
import json
import os

import numpy as np
import pytest
from PIL import Image

from conftest import GOLDEN, case_input, case_palette, fake_ffmpeg_tools

pytestmark = pytest.mark.gpu

with open(os.path.join(GOLDEN, "kat.json")) as _f:
    _KAT = json.load(_f)


@pytest.fixture(scope="module")
def d():
    import torch
    assert torch.cuda.is_available()
    from dither_pie_amd import dithering_lib
    return dithering_lib


@pytest.mark.parametrize("case", _KAT["cases"], ids=lambda c: c["name"])
def test_image_ditherer_matches_reference(d, orc, gold, case):
    arr = case_input(orc, case["input"])
    pal = case_palette(orc, case["palette"])
    it = d.ImageDitherer(len(pal), d.DitherMode(case["mode"]), pal, case["gamma"], dict(case["params"]))
    out = np.array(it.apply_dithering(Image.fromarray(arr)))
    assert out.dtype == np.uint8 and out.shape == arr.shape
    if case["full"]:
        assert np.array_equal(out, gold["out_" + case["name"]])
    assert orc.H(out) == case["h_out"]


def test_palette_none_uses_median_cut_and_keeps_it(d, orc, kat):
    for gamma in (False, True):
        ref = kat["misc"][f"auto_palette_gamma{int(gamma)}"]
        it = d.ImageDitherer(8, d.DitherMode.BAYER, None, gamma, {"size": "4x4"})
        out = np.array(it.apply_dithering(Image.fromarray(orc.rnd(48, 64, 17))))
        assert [list(map(int, c)) for c in it.palette] == ref["palette"]
        assert orc.H(out) == ref["h_out"]


def test_strategy_contract(d, orc):
    arr = orc.grad(37, 53)
    pal = np.array(orc.palr(40, 3), np.float32)
    px = arr.reshape(-1, 3).astype(np.float32)
    for strat, mode, params in [(d.NoDitherStrategy(), "none", {}), (d.BayerDitherStrategy("8x8"), "bayer", {"size": "8x8"}),
                                (d.InterleavedGradientNoiseDitherStrategy(2.0, 7), "IGN", {"scale": 2.0, "seed": 7}),
                                (d.BlueNoiseDitherStrategy(32, 1), "blue_noise", {"size": 32, "seed": 1}),
                                (d.ErrorDiffusionDitherStrategy("sierra", "true"), "error_diffusion",
                                 {"variant": "sierra", "serpentine": "true"}),
                                (d.MatrixDitherStrategy(np.linspace(0, 1, 35, dtype=np.float32).reshape(5, 7)), None, None)]:
        out = strat.dither(px, pal, (37, 53))
        assert out.shape == px.shape and out.dtype == np.float32
        if mode is not None:
            ref = orc.apply_dithering(arr, [tuple(int(v) for v in c) for c in pal], mode, params)
            assert np.array_equal(out.astype(np.uint8).reshape(arr.shape), ref), mode
        else:
            pf, oc, _ = orc.prepare_palette(pal.tolist(), False)
            ref = orc.ordered_u8(arr, pf, oc, None, "matrix", thr=strat.threshold_matrix)
            assert np.array_equal(out.astype(np.uint8).reshape(arr.shape), ref)
    with pytest.raises(ValueError):
        d.NoDitherStrategy().dither(px + 0.5, pal, (37, 53))
    # non-integer (linearised) palettes keep their float rows in the strategy-level API
    lin = np.array(orc.prepare_palette(orc.palr(16), True)[0])
    out = d.NoDitherStrategy().dither(px, lin, (37, 53))
    assert set(map(tuple, np.unique(out, axis=0))) <= set(map(tuple, lin))


def test_generate_blue_noise_and_ign_helpers(d, gold):
    assert np.array_equal(d.generate_blue_noise(32, 42), gold["blue_32_42"])
    s = d.InterleavedGradientNoiseDitherStrategy(2.5, 17)
    assert np.array_equal(s._generate_thresholds((37, 53)), gold["ign_37x53_s25_seed17"])


def test_frames_api_tiles_and_pickle(d, orc):
    import pickle
    import torch
    pal = orc.palr(64, 9)
    it = pickle.loads(pickle.dumps(d.ImageDitherer(64, d.DitherMode.BLUE_NOISE, pal, False, {"size": 32, "seed": 3})))
    frames = np.stack([orc.rnd(50, 70, s) for s in range(4)])
    out = it.apply_dithering_frames(torch.from_numpy(frames).cuda()).cpu().numpy()
    for i in range(4):
        assert np.array_equal(out[i], orc.apply_dithering(frames[i], pal, "blue_noise", {"size": 32, "seed": 3}))
    big = orc.grad(90, 64)
    full = orc.apply_dithering(big, pal, "blue_noise", {"size": 32, "seed": 3})
    from dither_pie_amd import sharding
    for lo, hi in sharding.row_bands(90, 4):
        band = sharding.dither_band(it, torch.from_numpy(np.ascontiguousarray(big[lo:hi])).cuda(), lo)
        assert np.array_equal(band.cpu().numpy(), full[lo:hi])
    with pytest.raises(ValueError):
        d.ImageDitherer(16, d.DitherMode.ERROR_DIFFUSION, pal).apply_dithering_frames(torch.from_numpy(big).cuda(), y0=4)
    with pytest.raises(NotImplementedError):
        d.ImageDitherer(16, d.DitherMode.HALFTONE, pal).apply_dithering(Image.fromarray(big))


def test_too_many_colours_is_reported(d, orc):
    from dither_pie_amd import DitherPieError
    pal = orc.palr(1100, 1)
    with pytest.raises(DitherPieError):
        d.ImageDitherer(1100, d.DitherMode.NONE, pal).apply_dithering(Image.fromarray(orc.rnd(8, 8, 1)))


def test_kmeans_gpu_matches_sklearn_fixture_and_oracle(d, orc, gold, kat):
    import torch
    from dither_pie_amd import kmeans
    for nm in ("km8", "km16", "km32"):
        m = kat["misc"][nm]
        arr = orc.rnd(m["h"], m["w"], m["seed"]) if m["kind"] == "rnd" else orc.grad(m["h"], m["w"])
        px = arr.reshape(-1, 3)
        init = px[gold[f"{nm}_init_idx"]].astype(np.float64)
        centers, inertia, n_iter = kmeans.lloyd(torch.from_numpy(px).cuda(), init)
        assert np.abs(centers - gold[f"{nm}_centers"]).max() < 1e-6
        assert abs(inertia - m["inertia"]) <= 1e-6 * m["inertia"]
        c_or, i_or, n_or = orc.kmeans_lloyd(px, init)
        assert np.abs(centers - c_or).max() < 1e-9
        diff = np.abs(centers.astype(int) - gold[f"{nm}_palette"])
        assert diff.max() <= 1 and (diff > 0).mean() <= 0.05


def test_generate_kmeans_palette_equals_reference_on_small_images(d, orc, gold, kat):
    """ColorReducer.generate_kmeans_palette against the REFERENCE's own palettes (dithering_lib.py:1845-1857) on the 11
    images of <= 10 000 pixels in tests/golden, where the reference is deterministic: device seeding (sklearn's draw
    order), Lloyd over the pixels with sklearn's labelling of equidistant pixels, astype(int).  Equal up to A.6's rounding
    noise at integer boundaries (+-1 on <= 5 % of the entries; in fact equal)."""
    import torch
    from conftest import case_input
    from dither_pie_amd import kmeans
    cases = []
    for nm in ("km8", "km16", "km32"):
        m = kat["misc"][nm]
        cases.append((nm, ["rnd", m["h"], m["w"], m["seed"]] if m["kind"] == "rnd" else ["grad", m["h"], m["w"]], m["K"], 42, m))
    for nm, m in sorted(kat["misc"]["kmeans_extra"].items()):
        cases.append((nm, m["input"], m["K"], m["random_state"], m))
    assert len(cases) == 11
    for nm, spec, K, rs, m in cases:
        arr = case_input(orc, spec)
        pal = d.ColorReducer.generate_kmeans_palette(Image.fromarray(arr), K, random_state=rs)
        assert len(pal) == K and all(isinstance(v, int) for c in pal for v in c)
        diff = np.abs(np.array(pal) - gold[f"{nm}_palette"])
        assert diff.max() <= 1 and (diff > 0).mean() <= 0.05, (nm, int(diff.max()))
        _, centers, inertia, n_iter = kmeans.fit_palette(torch.from_numpy(arr).cuda().reshape(-1, 3), K, rs)
        assert n_iter == m["n_iter"], nm
        assert np.abs(centers - gold[f"{nm}_centers"]).max() < 1e-9, nm
        assert abs(inertia - m["inertia"]) <= 1e-9 * m["inertia"], nm


def test_generate_kmeans_palette_on_few_colour_images(d, orc, gold, kat):
    """The product on the kmf_* fixtures (few colours: cluster means that are exact integers; the reference itself returns
    colour or colour - 1 there, differently from run to run -- tests/test_oracle_golden.py::
    test_kmeans_few_colour_images_integer_means): the oracle's palette exactly, hence 0 <= ours - reference <= 1 against every
    palette the reference produced."""
    for nm, m in sorted(kat["misc"]["kmeans_few"].items()):
        px = orc.few_colour_pixels(m["n"], m["colours"], m["seed"])
        pal = np.array(d.ColorReducer.generate_kmeans_palette(Image.fromarray(px.reshape(-1, 1, 3)), m["K"], random_state=42))
        centers, _, _ = orc.kmeans_lloyd(px, orc.kmeans_plusplus(px, m["K"], np.random.RandomState(42)))
        assert np.array_equal(pal, centers.astype(int)), nm
        for ref in gold[f"{nm}_palettes"]:
            assert (pal - ref).min() >= 0 and (pal - ref).max() <= 1, nm


def test_pixelized_example_end_to_end(d, orc, kat, gold):
    """examples/image_pixelized.json as dither_cli.process_single_image runs it (dither_cli.py:516, 423, 546-566) - the
    reference's one shipped k-means configuration - on a synthetic 400x300 stand-in for the absent test_300.png: regular
    pixelization to max_size 64 (86x64 = 5504 pixels: the deterministic regime), k-means 16 with random_state 42 of the
    pixelized image, error diffusion with its defaults, use_gamma, final NEAREST resize x8.  Palette and every output
    byte equal the reference's."""
    from dither_pie_amd import video_processor as v
    m = kat["misc"]["pixelized_example"]
    src = Image.fromarray(orc.imgl(*m["input"][1:]))
    small = v.pixelize_regular(src, m["max_size"])
    assert list(small.size) == m["small_size"] and orc.H(np.array(small)) == m["h_small"]
    pal = d.ColorReducer.generate_kmeans_palette(small, m["K"], random_state=42)
    assert [list(c) for c in pal] == m["palette"]
    out = d.ImageDitherer(m["K"], d.DitherMode("error_diffusion"), pal, True, {}).apply_dithering(small)
    assert np.array_equal(np.array(out), gold["pixelized_example_out"])
    big = out.resize((out.width * m["multiplier"], out.height * m["multiplier"]), Image.Resampling.NEAREST)
    assert orc.H(np.array(big)) == m["h_big"]


def test_lloyd_rejects_centres_outside_the_cube(d):
    import torch
    from dither_pie_amd import kmeans
    px = torch.zeros((64, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):
        kmeans.lloyd(px, np.array([[0.0, 0.0, 0.0], [300.0, 1.0, 1.0]]))
    with pytest.raises(ValueError):
        kmeans.lloyd(px, np.array([[0.0, -1.0, 0.0]]))


def test_generate_kmeans_palette_quality_and_determinism(d, orc):
    """A.6 (ii): inertia over the full image <= 1.01 x the inertia of an sklearn-style fit on a 10k sample."""
    arr = np.concatenate([orc.grad(150, 200), orc.rnd(150, 200, 3) // 2 + 60], axis=0)
    img = Image.fromarray(arr)
    p1 = d.ColorReducer.generate_kmeans_palette(img, 16)
    p2 = d.ColorReducer.generate_kmeans_palette(img, 16, random_state=42)
    assert p1 == p2 and len(p1) == 16 and all(len(c) == 3 and all(isinstance(v, int) for v in c) for c in p1)
    px = arr.reshape(-1, 3)

    def inertia(pal):
        P = np.array(pal, np.float64)
        return float((((px[:, None, :].astype(np.float64) - P[None]) ** 2).sum(2)).min(1).sum())

    rs = np.random.RandomState(0)
    sample = px[rs.choice(len(px), 10000, replace=False)]
    c_ref, _, _ = orc.kmeans_lloyd(sample, orc.kmeans_plusplus(sample, 16, np.random.RandomState(42)))
    assert inertia(p1) <= 1.01 * inertia(c_ref.astype(int))


def test_video_frame_pipeline(d, orc, gold, tmp_path):
    import torch
    from dither_pie_amd import video_processor as v
    img = Image.fromarray(orc.rnd(37, 53, 5))
    assert np.array_equal(np.array(v.pixelize_regular(img, 16)), gold["pixelize_regular_37x53_to16"])
    assert np.array_equal(np.array(v._apply_final_resize_to_frame(img, 3)), gold["final_resize_37x53_x3"])
    pal = orc.generate_uniform_palette(16)
    it = d.ImageDitherer(16, d.DitherMode.BAYER, pal, False, {"size": "4x4"})
    frames = np.stack([orc.rnd(60, 80, s) for s in range(3)])
    out = v.process_frames(torch.from_numpy(frames).cuda(), it, "regular", 16, 2).cpu().numpy()
    for i in range(3):
        small = np.array(Image.fromarray(frames[i]).resize(v._even_dimensions(80, 60, 16), Image.NEAREST))
        dith = orc.apply_dithering(small, pal, "bayer", {"size": "4x4"})
        ref = np.array(Image.fromarray(dith).resize(v._final_size(dith.shape[1], dith.shape[0], 2), Image.NEAREST))
        assert np.array_equal(out[i], ref)
    # the PNG worker, in place
    f = tmp_path / "frame_00001.png"
    Image.fromarray(frames[0]).save(f)
    assert v._process_single_frame(f, it, "regular", 16, 2) is True
    assert np.array_equal(np.array(Image.open(f)), out[0])
    assert v._process_single_frame(tmp_path / "missing.png", it) is False
    # a batch through VideoProcessor's batch helper, with one unreadable file -> retried, then reported failed
    files = []
    for i in range(3):
        p = tmp_path / f"b_{i:05d}.png"
        Image.fromarray(frames[i]).save(p)
        files.append(p)
    bad = tmp_path / "b_00003.png"
    bad.write_bytes(b"not a png")
    failed = v.VideoProcessor()._process_batch(files + [bad], it, None, 64, None)
    assert failed == [bad]
    assert np.array_equal(np.array(Image.open(files[1])), orc.apply_dithering(frames[1], pal, "bayer", {"size": "4x4"}))


@pytest.mark.gpu
def test_video_streaming_through_rawvideo_pipes(d, orc, tmp_path, monkeypatch):
    """process_video_streaming with stand-ins for ffmpeg/ffprobe on PATH (there is no ffmpeg in the image): the
    decoder stand-in emits 11 synthetic 64x48 rgb24 frames, the encoder stand-in stores what it receives; the
    stored frames must equal process_frames() on the same input, including the partial last batch."""
    import stat
    import sys
    import torch
    from dither_pie_amd import video_processor as v
    n, w, h = 11, 64, 48
    rs = np.random.RandomState(5)
    frames = rs.randint(0, 256, (n, h, w, 3)).astype(np.uint8)
    fake_ffmpeg_tools(tmp_path, monkeypatch, frames)
    it = d.ImageDitherer(16, d.DitherMode.BAYER, orc.palr(16, 3), False, {"size": "4x4"})
    seen = []
    vp = v.VideoProcessor(progress_callback=lambda f_, m: seen.append(f_))
    out_path = tmp_path / "out.bin"
    assert vp.process_video_streaming(str(tmp_path / "in.mp4"), str(out_path), it, ("regular", 32), batch_size=4,
                                      final_resize_multiplier=2) is True
    blob = out_path.read_bytes()
    size, body = blob.split(b"\n", 1)
    ref = v.process_frames(torch.from_numpy(frames).cuda(), it, "regular", 32, 2).cpu().numpy()
    assert size.decode() == f"{ref.shape[2]}x{ref.shape[1]}"
    got = np.frombuffer(body, np.uint8).reshape(ref.shape)
    assert np.array_equal(got, ref)
    assert seen[0] == 0.0 and seen[-1] == 1.0 and all(b >= a for a, b in zip(seen, seen[1:]))


@pytest.mark.gpu
def test_video_pipes_survive_failing_frames_on_the_device_path(d, orc, tmp_path, monkeypatch):
    """The failure policy of the pipe path (video_processor.py:325-336, 53-96) around the REAL kernels: a ditherer that
    raises whenever a marked frame is in its batch.  The batch is retried frame by frame, the marked frames are replaced
    by the previous good output frame, every other frame equals process_frames(), the call returns True."""
    import torch
    from dither_pie_amd import video_processor as v
    n, w, h = 10, 64, 48
    frames = np.random.RandomState(6).randint(0, 255, (n, h, w, 3)).astype(np.uint8)
    bad = {2, 7}
    for i in range(n):
        frames[i, 0, 0, 0] = 255 if i in bad else 0
    fake_ffmpeg_tools(tmp_path, monkeypatch, frames)

    class Flaky(d.ImageDitherer):
        def apply_dithering_frames(self, x, *a, **k):
            if bool((x[:, 0, 0, 0] == 255).any()):
                raise RuntimeError("injected device-side failure")
            return super().apply_dithering_frames(x, *a, **k)

    it = Flaky(16, d.DitherMode.BAYER, orc.palr(16, 3), False, {"size": "4x4"})
    out_path = tmp_path / "out.bin"
    assert v.VideoProcessor().process_video_streaming(str(tmp_path / "in.mp4"), str(out_path), it, None, batch_size=4) is True
    size, body = out_path.read_bytes().split(b"\n", 1)
    good = d.ImageDitherer(16, d.DitherMode.BAYER, orc.palr(16, 3), False, {"size": "4x4"})
    ref = v.process_frames(torch.from_numpy(frames).cuda(), good, None, 64, None).cpu().numpy()
    got = np.frombuffer(body, np.uint8).reshape(ref.shape)
    for i in range(n):
        assert np.array_equal(got[i], ref[i - 1] if i in bad else ref[i]), i


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, [0] * 8])
def test_overlapped_pipe_path_on_the_device_equals_the_serial_loop(d, orc, tmp_path, monkeypatch, devices):
    """SURVEY 8(f) rank 4 for real: reader thread / GPU stream / writer thread on rotating pinned slots against the serial
    loop, around the REAL kernels (regular pixelization -> Bayer -> x2 resize), 47 frames in batches of 4 (devices=None) and
    through eight worker streams of sharding.process_on_devices on the one GPU (devices=[0]*8: the in-process form of an
    8-GPU node; 32-frame batches): the encoder receives the same bytes, and they are process_frames() of the input."""
    import torch
    from dither_pie_amd import backend, video_processor as v
    n, w, h = 47, 64, 48
    frames = np.random.RandomState(8).randint(0, 256, (n, h, w, 3)).astype(np.uint8)
    fake_ffmpeg_tools(tmp_path, monkeypatch, frames)
    it = d.ImageDitherer(16, d.DitherMode.BAYER, orc.palr(16, 3), False, {"size": "4x4"})
    ref = v.process_frames(torch.from_numpy(frames).cuda(), it, "regular", 32, 2).cpu().numpy()
    assert v.output_size(h, w, "regular", 32, 2) == ref.shape[1:3]
    blobs = []
    for overlap in (False, True, True):
        vp = v.VideoProcessor(devices=devices)
        out_path = tmp_path / "out.bin"
        info = vp.get_video_info("in.mp4")
        assert vp._stream_through_pipes("in.mp4", str(out_path), it, "regular", 32, 4, 2, info, overlap=overlap) == n
        assert vp.last_pipe_stats["mode"] == ("overlapped" if overlap else "serial")
        blobs.append(out_path.read_bytes())
    assert blobs[0] == blobs[1] == blobs[2]
    size, body = blobs[0].split(b"\n", 1)
    assert size.decode() == f"{ref.shape[2]}x{ref.shape[1]}"
    assert np.array_equal(np.frombuffer(body, np.uint8).reshape(ref.shape), ref)
    if devices is not None:
        # eight worker streams + the default stream are nine live (device, stream) keys: none of them may have been evicted
        # and re-allocated per batch (round 4 kept 8)
        assert backend._ws_keep() >= 9
        assert len(backend._ws_cache) <= backend._ws_keep()
        assert all(e[2] == 0 for e in backend._ws_cache.values())


@pytest.mark.gpu
def test_spliced_slots_are_not_reused_before_a_slow_encoder_has_read_them(d, orc, tmp_path, monkeypatch):
    """The overlapped path hands its output slots to the encoder pipe by reference (vmsplice): a slot goes back into rotation only
    once the encoder is past its pages.  An encoder that sleeps before every 100 KB while 60 batches of 36 KB queue up behind it
    would receive overwritten frames if a slot came back early; the bytes it stores must be process_frames() of the input."""
    import torch
    from dither_pie_amd import video_processor as v
    n, w, h = 240, 64, 48
    frames = np.random.RandomState(9).randint(0, 256, (n, h, w, 3)).astype(np.uint8)
    fake_ffmpeg_tools(tmp_path, monkeypatch, frames, encoder_sleeps=0.02)
    it = d.ImageDitherer(16, d.DitherMode.BAYER, orc.palr(16, 3), False, {"size": "4x4"})
    vp = v.VideoProcessor(devices=[0])
    out_path = tmp_path / "out.bin"
    assert vp._stream_through_pipes("in.mp4", str(out_path), it, None, 64, 4, None, vp.get_video_info("in.mp4")) == n
    ref = v.process_frames(torch.from_numpy(frames).cuda(), it, None, 64, None).cpu().numpy()
    size, body = out_path.read_bytes().split(b"\n", 1)
    assert np.array_equal(np.frombuffer(body, np.uint8).reshape(ref.shape), ref)
    if v._PipeSplicer().ok:
        assert vp.last_pipe_stats.get("spliced_batches", 0) == n // 4


@pytest.mark.gpu
def test_workspace_cache_never_drops_an_entry_in_use(d):
    """backend._Launch: the per-(device, stream) scratch entry carries the lock that serialises that stream's launch
    sequences; while a launch is inside (or queued for) an entry, eviction passes it over however many other streams
    come by -- a fresh entry for the same key would hand a second thread a second lock."""
    import torch
    from dither_pie_amd import backend
    dev = torch.device("cuda", 0)
    backend.release_workspaces()
    keep = backend._ws_keep()
    with backend._Launch(dev, 1 << 20):
        held_key = (0, torch.cuda.current_stream(dev).cuda_stream)
        held = backend._ws_cache[held_key]
        streams = [torch.cuda.Stream() for _ in range(keep + 5)]
        for st in streams:
            with torch.cuda.stream(st), backend._Launch(dev, 4096):
                pass
        assert backend._ws_cache.get(held_key) is held and held[2] == 1
        assert len(backend._ws_cache) <= keep + 1
    assert held[2] == 0
    backend.release_workspaces()
    assert len(backend._ws_cache) == 0


def _first_occurrences(arr):
    packed = (arr[:, 0].astype(np.uint32) << 16) | (arr[:, 1].astype(np.uint32) << 8) | arr[:, 2]
    _, first = np.unique(packed, return_index=True)
    return arr[np.sort(first)]


def test_distinct_colours_on_device_match_numpy():
    """dp_distinct_first_u8 (the device side of the reference's set(image.getdata()), dithering_lib.py:1835-1843): the distinct
    colours in order of first occurrence equal numpy's -- noise, few colours, runs, one flat colour, counts around the block
    and window sizes of the kernels, one pixel, an unaligned buffer -- and ColorReducer._distinct_in_order takes that route for
    big images (the set built from it, and so the median cut, depends on the order)."""
    import torch
    from dither_pie_amd import backend as be
    from dither_pie_amd.dithering_lib import ColorReducer
    rs = np.random.RandomState(3)
    for n, top in ((150_000, 256), (400_000, 12), (120_001, 40)):
        arr = rs.randint(0, top, (n, 3)).astype(np.uint8)
        assert np.array_equal(ColorReducer._distinct_in_order(arr), _first_occurrences(arr))
    cases = [rs.randint(0, 256, (n, 3)).astype(np.uint8) for n in (1, 3, 255, 256, 2047, 2048, 2049, 4097, 1_000_003)]
    cases.append(np.tile(np.array([[9, 200, 31]], np.uint8), (300_000, 1)))                                   # flat
    cases.append(np.repeat(rs.randint(0, 256, (9000, 3)).astype(np.uint8), 37, axis=0))                        # runs
    cases.append(rs.randint(0, 256, (7, 3)).astype(np.uint8)[rs.randint(0, 7, 500_001)])                       # few colours
    cases.append(orc_image_like(rs))
    for arr in cases:
        t = torch.from_numpy(np.ascontiguousarray(arr)).cuda()
        assert np.array_equal(be.distinct_first(t).cpu().numpy(), _first_occurrences(arr)), len(arr)
    arr = cases[8]
    raw = torch.empty(3 * len(arr) + 3, dtype=torch.uint8, device="cuda")
    raw[3:] = torch.from_numpy(arr).cuda().reshape(-1)
    view = raw[3:].view(-1, 3)
    assert view.data_ptr() % 4 != 0
    assert np.array_equal(be.distinct_first(view).cpu().numpy(), _first_occurrences(arr))
    assert be.distinct_first(torch.empty((0, 3), dtype=torch.uint8, device="cuda")).shape[0] == 0
    assert torch.cuda.is_available()


def orc_image_like(rs):
    yy, xx = np.mgrid[0:540, 0:960]
    img = np.clip(np.stack([80 + 60 * np.sin(xx / 300.0) + 40 * (yy / 540.0), 110 + 50 * np.cos(yy / 200.0) + 20 * np.sin(xx / 97.0),
                            160 + 70 * (yy / 540.0) + 10 * np.sin((xx + yy) / 50.0)], -1) + rs.normal(0, 3, (540, 960, 3)), 0, 255)
    return img.astype(np.uint8).reshape(-1, 3)


@pytest.mark.gpu
def test_release_library_on_the_gpu(orc, tmp_path):
    """The product library (libditherpie_hip.so, no experiment switches) in a process of its own, as a user's process loads
    it: DP_* variables in the environment change nothing, the C2-shaped, the crowded-palette and the diffusion results equal
    the oracle's."""
    import subprocess
    import sys
    from conftest import ROOT
    code = """
import sys, numpy as np, torch
sys.path.insert(0, %r)
from dither_pie_amd import _lib
assert not _lib.EXPERIMENTS and _lib.LIB_PATH.endswith('libditherpie_hip.so')
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
from oracle import oracle as orc
from PIL import Image
arr = orc.rnd(270, 481, 5)
pal = orc.palr(256)
d = ImageDitherer(256, DitherMode.BAYER, pal, False, {'size': '8x8'}).prepare()
out = d.apply_dithering_frames(torch.from_numpy(arr).cuda()).cpu().numpy()
assert np.array_equal(out, orc.apply_dithering(arr, pal, 'bayer', {'size': '8x8'}))
img = orc.imgl(240, 320, 6, 'smooth')
mc = ColorReducer.reduce_colors(Image.fromarray(img), 256)
d = ImageDitherer(256, DitherMode.BAYER, mc, False, {'size': '8x8'}).prepare()
out = d.apply_dithering_frames(torch.from_numpy(img).cuda()).cpu().numpy()
assert np.array_equal(out, orc.apply_dithering(img, mc, 'bayer', {'size': '8x8'}))
u16 = orc.generate_uniform_palette(16)
d = ImageDitherer(16, DitherMode.ERROR_DIFFUSION, u16, False, {'variant': 'floyd_steinberg'})
out = d.apply_dithering_frames(torch.from_numpy(arr[:64, :96].copy()).cuda()).cpu().numpy()
assert np.array_equal(out, orc.apply_dithering(arr[:64, :96].copy(), u16, 'error_diffusion', {'variant': 'floyd_steinberg', 'serpentine': 'false'}))
print('release ok')
""" % ROOT
    env = {k: v for k, v in os.environ.items() if k != "DITHER_PIE_EXPERIMENTS"}
    env.update(DP_NO_COMPACT_KERNEL="1", DP_KMEANS_CELLS="0", DP_FORCE_TABLE="u8", DP_ED_TEST_GIVEUP="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "release ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.gpu
def test_bench_three_rank_rehearsal_uneven_shards():
    """A wider rehearsal of the driver's 8-GPU SCALE run than two ranks, inside what a one-GPU box allows (its process guard
    admits six processes on the card; this process, the elastic agent and the ranks all count -- five ranks were killed by it):
    bench.py --gpus 3 --rehearse-on-one-gpu, self-launched through torch.distributed.run.  Three does not divide C5's 1000
    frames: rank 0 takes 333 of them in four nearly equal launches (83, 83, 83, 84) and a 1440-row band of C4's 8K image; the
    collectives run over gloo; one JSON line.  What the real run adds is RCCL itself, nothing else."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    env.pop("DITHER_PIE_EXPERIMENTS", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--rehearse-on-one-gpu", "--steps", "2",
                        "--warmup", "1", "--frames", "2", "--no-cpu-baseline"], capture_output=True, text=True, env=env,
                       timeout=1100, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res["n_gpus"] == 3 and res["rccl_world_size"] == 3 and "rehearsal" in res
    assert res["parity_kat_4k"] is True and res["scaling"] == "weak"
    c5 = res["c5_video"]
    assert c5["frames_this_rank"] == 333 and c5["n_gpus"] == 3 and c5["scaling"] == "strong"
    assert c5["launches_per_pass"] == 4 and c5["frames_per_launch"] == [83, 83, 83, 84] and c5["frames_per_s"] > 0
    assert "3 band(s)" in res["c4_kmeans_pass"]["workload"] and res["c4_kmeans_pass"]["iterations_of_the_fit"] > 0
    assert "c5_pipes" not in res and "cpu_baseline" not in res   # host-path legs: one rank only
    assert res["roofline"]["frac"] > 0 and res["value"] > 0


@pytest.mark.gpu
def test_bench_two_rank_rehearsal():
    """bench.py's N > 1 path end to end, as the driver's SCALE run will start it, rehearsed on the one GPU of the box: the
    parent starts two ranks through torch.distributed.run before it touches the GPU, both ranks use cuda:0, the collectives
    run over gloo.  Exactly one JSON line on stdout, carrying the per-rank shares of C5 (500 of the 1000 frames) and C4 (two
    row bands), the C2 known answer, and the rehearsal mark (the numbers of such a run mean nothing)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    env.pop("DITHER_PIE_EXPERIMENTS", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "2",
                        "--warmup", "1", "--frames", "2", "--no-cpu-baseline"], capture_output=True, text=True, env=env,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["rccl_world_size"] == 2 and "rehearsal" in res
    assert res["steps"] == 2 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["parity_kat_4k"] is True
    assert res["c5_video"]["frames_this_rank"] == 500 and res["c5_video"]["n_gpus"] == 2 and res["c5_video"]["scaling"] == "strong"
    assert "2 band(s)" in res["c4_kmeans_pass"]["workload"]
    assert "cpu_baseline" not in res
    assert res["roofline"]["frac"] > 0 and res["value"] > 0
