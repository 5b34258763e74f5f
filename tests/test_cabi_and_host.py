"""CPU tier: the C-ABI library loads and exports every symbol include/ditherpie_hip.h declares (no compute
calls), and the host-side logic of the drop-in mirror matches the golden fixtures / the oracle."""
import ctypes as C
import os
import sys
import pickle
import re

import numpy as np
import pytest

from conftest import ROOT, fake_ffmpeg_tools as _fake_ffmpeg_tools


@pytest.fixture(scope="module")
def lib():
    import dither_pie_amd
    if not os.path.exists(dither_pie_amd._lib.LIB_PATH):
        dither_pie_amd.build()
    return dither_pie_amd.load()


def _header_functions():
    src = open(os.path.join(ROOT, "include", "ditherpie_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    from dither_pie_amd import _lib
    declared = _header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in the header but not exported"
    assert sorted(_lib.EXPORTS) == declared, "ctypes signature table and header disagree"
    assert lib.dp_version() >= 100


def test_error_reporting_without_gpu(lib):
    # argument errors are detected before any HIP call
    from dither_pie_amd._lib import DP_EINVAL
    assert lib.dp_palette_info(None, None, None, None) == DP_EINVAL
    assert b"NULL" in lib.dp_last_error()
    assert lib.dp_ordered_workspace_bytes(24, 2160, 3840) >= 24 * 2160 * 3840 // 8


def test_product_kdtree_build_matches_golden_and_scipy(lib, gold, kat):
    def build(P):
        P = np.ascontiguousarray(P, np.float64)
        K = len(P)
        n = 2 * K + 2
        idx = np.zeros(K, np.int32)
        sd, st, en, le, gr = (np.zeros(n, np.int32) for _ in range(5))
        sp = np.zeros(n, np.float64)
        nn = C.c_int()
        rc = lib.dp_kdtree_build_host(P.ctypes.data, K, idx.ctypes.data, sd.ctypes.data, sp.ctypes.data, st.ctypes.data,
                                      en.ctypes.data, le.ctypes.data, gr.ctypes.data, C.byref(nn))
        assert rc == 0
        m = nn.value
        return idx, sd[:m], sp[:m], st[:m], en[:m], le[:m], gr[:m]

    for nm in kat["misc"]["tree_palettes"]:
        idx, sd, sp, st, en, le, gr = build(gold[f"tree_{nm}_pts"])
        nodes = gold[f"tree_{nm}_nodes"]
        assert np.array_equal(idx, gold[f"tree_{nm}_indices"]), nm
        assert np.array_equal(sd, nodes[:, 0]) and np.array_equal(st, nodes[:, 1]) and np.array_equal(en, nodes[:, 2])
        assert np.array_equal(le, nodes[:, 3]) and np.array_equal(gr, nodes[:, 4])
        inner = nodes[:, 0] >= 0
        assert np.array_equal(sp[inner], gold[f"tree_{nm}_splits"][inner])
    sp_spatial = pytest.importorskip("scipy.spatial")
    rs = np.random.RandomState(11)
    for trial in range(40):
        K = int(rs.randint(1, 257))
        P = (rs.randint(0, 256, (K, 3)) if trial % 2 else rs.randint(0, 6, (K, 3)) * 51).astype(np.float64)
        assert np.array_equal(build(P)[0], np.asarray(sp_spatial.cKDTree(P, leafsize=10).indices))


def test_enums_tables_and_tap_tables(orc, gold, kat):
    from dither_pie_amd import dithering_lib as d
    assert {m.name: m.value for m in d.DitherMode} == kat["misc"]["dither_modes"]
    assert d.DitherMode("IGN") is d.DitherMode.INTERLEAVED_GRADIENT_NOISE
    for name in ["BAYER2x2", "BAYER4x4", "BAYER8x8", "BAYER16x16", "PSX4x4"]:
        m = getattr(d.DitherUtils, name)
        assert m.dtype == np.float32 and np.array_equal(m, gold["table_" + name])
    assert np.array_equal(d.DitherUtils.get_threshold_matrix(d.DitherMode.NONE), np.ones((1, 1), np.float32))
    assert np.array_equal(d.PolkaDotDitherStrategy(8, 1.5).threshold_matrix, gold["polka_8_15"])
    assert np.array_equal(d.PolkaDotDitherStrategy(5, 0.7).threshold_matrix, gold["polka_5_07"])
    assert np.array_equal(orc.polka_dot_matrix(5, 0.7), gold["polka_5_07"])
    with pytest.raises(ValueError):
        d.DitherUtils.get_threshold_matrix(d.DitherMode.HALFTONE)
    for k, (taps, div) in orc.ED_KERNELS.items():
        kk = d.ErrorDiffusionKernel.get_kernel(k)
        assert kk["weights"] == taps and kk["divisor"] == div
    assert d.ErrorDiffusionKernel.get_kernel("nope") is d.ErrorDiffusionKernel.FLOYD_STEINBERG
    assert d.ErrorDiffusionKernel.list_kernels() == list(orc.ED_KERNELS)


def test_parameter_metadata_matches_reference(kat):
    from dither_pie_amd import dithering_lib as d
    ref = kat["misc"]["mode_parameters"]
    for mode in (d.DitherMode.BAYER, d.DitherMode.BLUE_NOISE, d.DitherMode.INTERLEAVED_GRADIENT_NOISE,
                 d.DitherMode.ERROR_DIFFUSION, d.DitherMode.POLKA_DOT, d.DitherMode.HYBRID,
                 d.DitherMode.ADAPTIVE_VARIANCE, d.DitherMode.OSTROMOUKHOV):
        assert d.ImageDitherer.get_mode_parameters(mode) == ref[mode.value]
        assert d.ImageDitherer.mode_has_parameters(mode)
    assert d.ImageDitherer.get_mode_parameters(d.DitherMode.NONE) is None and ref["none"] is None
    assert d.ImageDitherer.get_mode_parameters(d.DitherMode.PERCEPTUAL) is None and ref["perceptual"] is None
    assert d.OstromoukhovDitherStrategy.COEFFS_TABLE[11] == (501, 224, 211)


def test_strategy_construction_and_errors():
    from dither_pie_amd import dithering_lib as d
    it = d.ImageDitherer(16, d.DitherMode.BAYER, [(0, 0, 0), (255, 255, 255)], False, {"size": "8x8"})
    s = it._get_dither_strategy(it.dither_mode)
    assert isinstance(s, d.BayerDitherStrategy) and s.get_current_parameters() == {"size": "8x8"}
    s = d.ImageDitherer(dither_mode=d.DitherMode.ERROR_DIFFUSION)._get_dither_strategy(d.DitherMode.ERROR_DIFFUSION)
    assert s.get_current_parameters() == {"variant": "atkinson", "serpentine": "false"}
    with pytest.raises(TypeError):  # unknown parameter -> TypeError from the constructor, as in the reference
        d.ImageDitherer(dither_params={"bogus": 1})._get_dither_strategy(d.DitherMode.BAYER)
    with pytest.raises(ValueError):
        d.ImageDitherer()._get_dither_strategy("not a mode")
    with pytest.raises(NotImplementedError):
        d.ImageDitherer()._get_dither_strategy(d.DitherMode.HALFTONE)
    blob = pickle.dumps(it)
    assert len(blob) < 600 and pickle.loads(blob).dither_params == {"size": "8x8"}


def test_palette_preparation_and_gamma_tables(orc, gold):
    from dither_pie_amd import _tables, dithering_lib as d
    assert np.array_equal(_tables.LUT_IN, gold["lut_in"]) and np.array_equal(_tables.LUT_OUT, gold["lut_out"])
    assert np.array_equal(_tables.PAL_LIN, gold["pal_lin_table"])
    for pal in (orc.palr(256), orc.generate_uniform_palette(16), [(0, 0, 0)]):
        for gamma in (False, True):
            a, b = d.prepare_palette(pal, gamma), orc.prepare_palette(pal, gamma)
            for x, y in zip(a, b):
                assert (x is None and y is None) or np.array_equal(x, y)


def test_palette_producers(orc, gold, kat):
    from PIL import Image
    from dither_pie_amd.dithering_lib import ColorReducer
    for n in (0, 1, 2, 8, 16, 27, 64, 100, 256):
        assert ColorReducer.generate_uniform_palette(n) == orc.generate_uniform_palette(n)
    for n in (2, 8, 16, 27, 64, 256):
        assert np.array_equal(np.array(ColorReducer.generate_uniform_palette(n), np.int32), gold[f"uniform_{n}"])
    imgs = {"rnd40x50": orc.rnd(40, 50, 41), "grad64x96": orc.grad(64, 96)}
    for key, ref in kat["misc"]["median_cut"].items():
        nm, n = key.rsplit("_", 1)
        if nm.startswith("imgl:"):  # image-like content of the golden cases with palettes extracted from the image
            _, hh, ww, seed, kind = nm.split(":")
            img = orc.imgl(int(hh), int(ww), int(seed), kind)
        elif nm.startswith("rnd:"):   # C1: examples/image_basic.json's median cut of rnd(512, 512, 1234)
            _, hh, ww, seed = nm.split(":")
            img = orc.rnd(int(hh), int(ww), int(seed))
        else:
            img = imgs[nm]
        got = ColorReducer.reduce_colors(Image.fromarray(img), int(n))
        assert [list(map(int, c)) for c in got] == ref, key
    assert ColorReducer.median_cut([], 3) == [(0, 0, 0)]


def _image_cases(seed, count):
    rs = np.random.RandomState(seed)
    for t in range(count):
        h, w = int(rs.randint(1, 90)), int(rs.randint(1, 120))
        kind = t % 4
        if kind == 0:
            a = rs.randint(0, 256, (h, w, 3))
        elif kind == 1:
            a = rs.randint(0, 6, (h, w, 3)) * 40  # few colours, many repeats
        elif kind == 2:
            y, x = np.mgrid[0:h, 0:w]
            a = np.stack([x % 256, (y * 3) % 256, ((x + y) // 2) % 256], -1) + rs.randint(0, 3, (h, w, 3))
        else:
            a = np.full((h, w, 3), rs.randint(0, 256))
        yield np.clip(a, 0, 255).astype(np.uint8)


def test_reduce_colors_matches_the_list_construct():
    """reduce_colors builds the set from the distinct colours only and cuts numpy arrays; the result has to be the one
    the reference's construct gives (dithering_lib.py:1835-1843: list(set(getdata())) then the list median cut),
    set iteration order included."""
    import math
    from PIL import Image
    from dither_pie_amd.dithering_lib import ColorReducer
    for a in _image_cases(7, 24):
        im = Image.fromarray(a, "RGB")
        flat = [tuple(int(v) for v in px) for px in a.reshape(-1, 3)]
        for k in (1, 2, 3, 16, 17, 256):
            depth = int(math.log2(k)) if k > 1 else 0
            assert ColorReducer.reduce_colors(im, k) == ColorReducer.median_cut(list(set(flat)), depth)


def test_video_helpers(kat):
    from dither_pie_amd import video_processor as v
    for a, b, c, ref in kat["misc"]["even_dims"]:
        assert list(v.NeuralPixelizer._compute_even_dimensions(a, b, c)) == ref
        assert list(v._even_dimensions(a, b, c)) == ref
    assert v._final_size(53, 37, 3) == (160, 112)
    seen = []
    vp = v.VideoProcessor(progress_callback=lambda f, m: seen.append((f, m)))
    assert vp.process_video_streaming("/nonexistent/in.mp4", "/tmp/out.mp4", None) is False  # no ffmpeg / no file
    assert seen[0][0] == 0.0 and seen[-1][0] == 1.0 and seen[-1][1].startswith("Error")
    info = vp.get_video_info("/nonexistent/in.mp4")
    assert info == {"fps": 30.0, "width": 1920, "height": 1080, "duration": None, "frame_count": None}


def test_fix_failed_frames_copies_nearest(tmp_path):
    from dither_pie_amd.video_processor import VideoProcessor
    files = []
    for i in range(5):
        f = tmp_path / f"frame_{i:05d}.png"
        f.write_bytes(bytes([i]) * 10)
        files.append(f)
    VideoProcessor()._fix_failed_frames([files[0], files[3]], files)
    assert files[0].read_bytes() == files[1].read_bytes() == bytes([1]) * 10  # no previous: take the next good one
    assert files[3].read_bytes() == bytes([2]) * 10                            # previous good frame


def test_sharding_helpers():
    from dither_pie_amd import sharding as s
    for n, world in [(1000, 8), (7, 3), (3, 8), (0, 2)]:
        blocks = [s.shard_range(n, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
        assert max(hi - lo for lo, hi in blocks) - min(hi - lo for lo, hi in blocks) <= 1
    assert s.row_bands(4320, 8) == [(i * 540, (i + 1) * 540) for i in range(8)]
    with pytest.raises(ValueError):
        s.shard_range(10, 3, 2)


def test_product_path_fails_loudly_without_gpu():
    import torch
    from dither_pie_amd import DitherPieError, backend
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(DitherPieError):
        backend.Palette(np.zeros((2, 3), np.float32), np.zeros((2, 3), np.uint8))


def test_product_never_imports_the_oracle():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import dither_pie_amd, dither_pie_amd.dithering_lib, "
            "dither_pie_amd.video_processor, dither_pie_amd.kmeans, dither_pie_amd.sharding, dither_pie_amd.backend; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'") % ROOT
    subprocess.check_call([sys.executable, "-c", code])
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dither_pie_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "from oracle" not in src and "import oracle" not in src and "dp_oracle" not in src, f


@pytest.mark.parametrize("bad,expect", [
    ({3}, [0, 1, 2, 2, 4, 5, 6, 7, 8, 9, 10]),                 # one frame: the previous good output frame
    ({0, 1}, [2, 2, 2, 3, 4, 5, 6, 7, 8, 9, 10]),              # nothing before them: the next good one
    ({0, 1, 2, 3, 4, 5}, [6, 6, 6, 6, 6, 6, 6, 7, 8, 9, 10]),  # a whole batch and more at the head of the video
    ({4, 5, 6, 7, 10}, [0, 1, 2, 3, 3, 3, 3, 3, 8, 9, 9]),     # a whole batch in the middle, and the last frame
])
def test_pipe_path_keeps_the_reference_failure_policy(tmp_path, monkeypatch, bad, expect):
    """video_processor.py:325-336, 53-96, 473-475 on the rawvideo-pipe path: a batch that raises is retried frame by
    frame (3 attempts), a frame that keeps failing is replaced by the nearest good OUTPUT frame (previous first) and the
    video continues with True.  The batch function is a stand-in that inverts a frame and raises on chosen ones (the
    device path is exercised in tests/test_gpu_api.py); frames carry their index so that substitutions are visible."""
    import torch
    from dither_pie_amd import video_processor as v
    n, h, w = 11, 6, 8
    frames = np.zeros((n, h, w, 3), np.uint8)
    for i in range(n):
        frames[i] = i
    _fake_ffmpeg_tools(tmp_path, monkeypatch, frames)
    attempts = {}

    def run(x):
        ids = [int(f[0, 0, 0]) for f in x]
        for i in ids:
            attempts[i] = attempts.get(i, 0) + 1
            if i in bad:
                raise RuntimeError(f"injected failure on frame {i}")
        return 255 - x.clone()

    seen = []
    vp = v.VideoProcessor(progress_callback=lambda f, m: seen.append((f, m)))
    info = vp.get_video_info("in.mp4")
    out_path = tmp_path / "out.bin"
    done = vp._stream_through_pipes("in.mp4", str(out_path), None, None, 64, 4, None, info, run=run)
    assert done == n
    size, body = out_path.read_bytes().split(b"\n", 1)
    assert size == f"{w}x{h}".encode()
    got = np.frombuffer(body, np.uint8).reshape(n, h, w, 3)
    assert [255 - int(f[0, 0, 0]) for f in got] == expect
    for i in bad:   # 1 try inside its batch (unless an earlier frame of the batch raised first) + 3 on its own
        assert attempts[i] in (v.VideoProcessor.ATTEMPTS, v.VideoProcessor.ATTEMPTS + 1)
    assert seen[-1][0] == 0.9


def test_pipe_path_fails_only_when_nothing_succeeds(tmp_path, monkeypatch):
    from dither_pie_amd import video_processor as v
    frames = np.zeros((5, 4, 4, 3), np.uint8)
    _fake_ffmpeg_tools(tmp_path, monkeypatch, frames)
    vp = v.VideoProcessor()

    def run(x):
        raise RuntimeError("always")

    with pytest.raises(RuntimeError, match="no frame"):
        vp._stream_through_pipes("in.mp4", str(tmp_path / "o.bin"), None, None, 64, 2, None, vp.get_video_info("in.mp4"), run=run)


def test_pipe_path_treats_a_dead_device_as_fatal(tmp_path, monkeypatch):
    """The reference's retry-and-substitute policy (video_processor.py:325-336) is about frames that fail -- PNG and I/O
    errors.  A failure of the DEVICE (the library's DP_EHIP / DP_ENOMEM, a HIP or out-of-memory error out of torch) is not a
    frame's fault: no frame-by-frame retries (3 x batch doomed launches), no video whose tail is copies of the last good
    frame reported as success -- the error leaves _stream_through_pipes, and process_video_streaming returns False."""
    from dither_pie_amd import video_processor as v
    from dither_pie_amd._lib import DP_EHIP, DP_EINVAL, DitherPieError
    frames = np.zeros((9, 4, 4, 3), np.uint8)
    for i in range(9):
        frames[i] = i
    _fake_ffmpeg_tools(tmp_path, monkeypatch, frames)
    calls = []

    def run(x):
        calls.append(len(x))
        if int(x[0, 0, 0, 0]) >= 4:          # the second batch: the GPU is gone
            raise DitherPieError(DP_EHIP, "hipErrorIllegalAddress")
        return 255 - x.clone()

    vp = v.VideoProcessor()
    with pytest.raises(DitherPieError):
        vp._stream_through_pipes("in.mp4", str(tmp_path / "o.bin"), None, None, 64, 4, None, vp.get_video_info("in.mp4"), run=run)
    assert calls == [4, 4]                    # the failing batch was tried once, no frame of it on its own
    assert v.VideoProcessor._device_is_gone(RuntimeError("HIP error: an illegal memory access was encountered"))
    assert v.VideoProcessor._device_is_gone(DitherPieError(4, "hipMalloc failed"))
    assert not v.VideoProcessor._device_is_gone(DitherPieError(DP_EINVAL, "bad argument"))
    assert not v.VideoProcessor._device_is_gone(ValueError("frames of one batch differ in size"))


@pytest.mark.parametrize("bad", [set(), {3}, {0, 1}, {4, 5, 6, 7, 10, 40}, set(range(9))])
def test_overlapped_pipe_path_writes_the_bytes_of_the_serial_loop(tmp_path, monkeypatch, bad):
    """The three concurrent stages on rotating batch slots (reader thread / submit / writer thread) against the serial loop
    of rounds 2-4, same stream, same batch function with failures injected: identical encoder input, byte for byte,
    including partial last batches and substituted frames; 41 frames in batches of 4 keep every slot in rotation."""
    from dither_pie_amd import video_processor as v
    n, h, w = 41, 6, 8
    frames = np.random.RandomState(3).randint(0, 256, (n, h, w, 3)).astype(np.uint8)
    for i in range(n):
        frames[i, 0, 0, 0] = i
    _fake_ffmpeg_tools(tmp_path, monkeypatch, frames)

    def run(x):
        for f in x:
            if int(f[0, 0, 0]) in bad:
                raise RuntimeError("injected")
        return 255 - x.clone()

    blobs = []
    for overlap in (False, True):
        vp = v.VideoProcessor()
        out_path = tmp_path / f"out_{int(overlap)}.bin"
        done = vp._stream_through_pipes("in.mp4", str(out_path), None, None, 64, 4, None, vp.get_video_info("in.mp4"), run=run,
                                        overlap=overlap)
        assert done == n
        st = vp.last_pipe_stats
        assert st["mode"] == ("overlapped" if overlap else "serial") and st["frames"] == n and st["wall_s"] > 0
        blobs.append(out_path.read_bytes())
    assert blobs[0] == blobs[1]
    assert len(blobs[0].split(b"\n", 1)[1]) == n * h * w * 3


def test_overlapped_pipe_path_ends_when_a_stage_fails(tmp_path, monkeypatch):
    """No stage may be left blocked on a queue or a pipe when another one fails: an encoder that dies mid-stream (the
    writer gets EPIPE), a decoder stream that is not a whole number of frames (the reader raises), a batch function that
    raises a device error (the submitting thread) -- each ends the call with the error, within seconds."""
    import threading
    import time
    from dither_pie_amd import video_processor as v
    from dither_pie_amd._lib import DP_EHIP, DitherPieError
    n, h, w = 64, 16, 16
    frames = np.zeros((n, h, w, 3), np.uint8)

    def call(run, **tools):
        d = tmp_path / f"t{len(list(tmp_path.iterdir()))}"
        d.mkdir()
        _fake_ffmpeg_tools(d, monkeypatch, frames, **tools)
        vp = v.VideoProcessor()
        box = {}

        def go():
            try:
                box["r"] = vp._stream_through_pipes("in.mp4", str(d / "o.bin"), None, None, 64, 4, None, vp.get_video_info("in.mp4"),
                                                    run=run)
            except BaseException as e:  # noqa: BLE001
                box["e"] = e
        t = threading.Thread(target=go, daemon=True)
        t0 = time.time()
        t.start()
        t.join(30)
        assert not t.is_alive(), "the pipe path hangs"
        assert time.time() - t0 < 30
        return box

    box = call(lambda x: x.clone(), encoder_dies_after=5 * h * w * 3)
    assert isinstance(box.get("e"), (BrokenPipeError, RuntimeError)), box        # EPIPE, or "ffmpeg failed (encoder 3)"
    box = call(lambda x: x.clone(), trailing=b"xyz")
    assert isinstance(box.get("e"), RuntimeError) and "whole number" in str(box["e"]), box

    def dead(x):
        if int(x.shape[0]) and dead.calls >= 3:
            raise DitherPieError(DP_EHIP, "hipErrorIllegalAddress")
        dead.calls += 1
        return x.clone()
    dead.calls = 0
    box = call(dead)
    assert isinstance(box.get("e"), DitherPieError), box
    # and the public entry point turns all of it into False, as the reference does (video_processor.py:386-390)
    d = tmp_path / "pub"
    d.mkdir()
    _fake_ffmpeg_tools(d, monkeypatch, frames, encoder_dies_after=100)
    assert v.VideoProcessor().process_video_streaming("in.mp4", str(d / "o.bin"), None) is False


def test_compiled_pipe_standins_and_a_long_stream(tmp_path, monkeypatch):
    """tools/pipe_standin.c (what bench.py's c5_pipes leg drives): decoder stream = tools/pipe_standin.frames(), the encoder's
    order-sensitive checksum and kept frames; 400 frames of 160x120 through the overlapped path in batches of 15."""
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pipe_standin as ps
    from dither_pie_amd import video_processor as v
    d = ps.build(str(tmp_path / "bin"))
    n, h, w = 400, 120, 160
    for k, val in ps.environment(d, n, h, w, distinct=5, keep=3).items():
        if k == "PATH" or k.startswith("DP_STANDIN_"):
            monkeypatch.setenv(k, val)
    vp = v.VideoProcessor()
    info = vp.get_video_info("in.mp4")
    assert (info["width"], info["height"], info["frame_count"], info["fps"]) == (w, h, n, 25.0)
    out = tmp_path / "o.bin"
    assert vp._stream_through_pipes("in.mp4", str(out), None, None, 64, 15, None, info, run=lambda x: 255 - x.clone()) == n
    summary, kept = ps.read_summary(str(out))
    src = ps.frames(n, h, w, distinct=5)
    assert [int(np.frombuffer(f[0, 0].tobytes() + f[0, 1, :1].tobytes(), np.uint32)[0]) for f in src[:7]] == list(range(7))
    assert summary["frames"] == n and summary["bytes"] == n * h * w * 3 and summary["keep"] == 3
    assert np.array_equal(kept, 255 - src[:3])
    assert summary["wsum"] == ps.weighted_sum(255 - src)
    assert vp.last_pipe_stats["mode"] == "overlapped" and vp.last_pipe_stats["frames"] == n


def test_pipe_splicer_hands_pages_to_a_pipe_and_knows_what_was_consumed(tmp_path):
    """_PipeSplicer (vmsplice): the bytes arrive, in order, next to ordinary writes on the same pipe; unread_bytes() is what the
    overlapped path's writer uses to decide when a spliced slot may be reused."""
    import subprocess
    from dither_pie_amd.video_processor import _PipeSplicer
    sp = _PipeSplicer()
    if not sp.ok:
        pytest.skip("no vmsplice here")
    a = np.random.RandomState(1).randint(0, 256, 3_000_001, dtype=np.uint8)
    out = tmp_path / "got.bin"
    p = subprocess.Popen(["sh", "-c", f"sleep 0.3; cat > {out}"], stdin=subprocess.PIPE, bufsize=0)
    fd = p.stdin.fileno()
    p.stdin.write(b"head")
    assert sp.splice_all(fd, a.ctypes.data, a.size) is True     # blocks until the last page is queued: the reader has started by then
    assert 0 <= _PipeSplicer.unread_bytes(fd) <= 1 << 20
    p.stdin.write(b"tail")
    p.stdin.close()
    assert p.wait() == 0
    assert out.read_bytes() == b"head" + a.tobytes() + b"tail"
    # a reader that is gone: BrokenPipeError, not a dead process
    q = subprocess.Popen(["true"], stdin=subprocess.PIPE, bufsize=0)
    q.wait()
    with pytest.raises(BrokenPipeError):
        sp.splice_all(q.stdin.fileno(), a.ctypes.data, a.size)


@pytest.mark.parametrize("rotation,swap", [(None, False), ("90", True), ("-90.000000", True), ("180", False), ("270", True)])
def test_pipe_path_follows_rotation_metadata(tmp_path, monkeypatch, rotation, swap):
    """Phone footage: ffprobe reports the CODED size plus a rotate tag / display matrix; ffmpeg rotates while decoding
    (the reference's plain extraction, video_processor.py:208-217, does), so the pipe must be sliced with the displayed
    geometry.  The decoder stand-in asserts it was asked for exactly that and was not told -noautorotate."""
    from dither_pie_amd import video_processor as v
    n, h, w = 3, 6, 10   # displayed
    frames = np.arange(n * h * w * 3, dtype=np.uint32).astype(np.uint8).reshape(n, h, w, 3)
    _fake_ffmpeg_tools(tmp_path, monkeypatch, frames, rotation=rotation, coded=(h, w) if swap else (w, h))
    vp = v.VideoProcessor()
    info = vp.get_video_info("in.mp4")
    assert (info["width"], info["height"]) == ((h, w) if swap else (w, h))
    out_path = tmp_path / "o.bin"
    assert vp._stream_through_pipes("in.mp4", str(out_path), None, None, 64, 2, None, info, run=lambda x: x.clone()) == n
    size, body = out_path.read_bytes().split(b"\n", 1)
    assert size == f"{w}x{h}".encode() and body == frames.tobytes()


def test_release_library_reads_no_environment():
    """csrc/Makefile builds the sources twice: libditherpie_hip.so (the product) must not import getenv nor carry the name of
    any DP_* experiment switch; libditherpie_hip_exp.so (-DDP_EXPERIMENTS: what the `switches` fixture of conftest.py maps for
    the tests that force a table / kernel / schedule) has them compiled in.  Both export the whole C ABI, and the session
    loads the product library."""
    import subprocess
    from dither_pie_amd import _lib
    here = os.path.dirname(_lib.__file__)
    rel, exp = os.path.join(here, "libditherpie_hip.so"), os.path.join(here, "libditherpie_hip_exp.so")
    assert os.path.exists(rel) and os.path.exists(exp)
    assert not _lib.EXPERIMENTS and _lib.LIB_PATH == rel
    sym = {p: subprocess.run(["nm", "-D", p], capture_output=True, text=True, check=True).stdout for p in (rel, exp)}
    assert "getenv" not in sym[rel] and "getenv" in sym[exp]
    names = {p: set(re.findall(rb"DP_[A-Z0-9_]{3,}", open(p, "rb").read())) for p in (rel, exp)}
    assert not (names[rel] - {b"DP_MODE_MATRIX"}), names[rel]   # (DP_MODE_MATRIX: an error message of the C ABI, not a switch)
    assert {b"DP_KMEANS_CELLS", b"DP_FORCE_TABLE", b"DP_NO_COMPACT_KERNEL"} <= names[exp]
    for p in (rel, exp):
        for fn in _lib.EXPORTS:
            assert re.search(rf"\bT {fn}\b", sym[p]), (p, fn)


def test_native_median_cut_equals_the_python_path(kat):
    """ColorReducer.reduce_colors through dp_median_cut_host (CPython's set order replayed natively + counting-sort cut)
    against the same function with real Python sets and numpy sorts -- which the reference fixtures pin
    (test_median_cut_matches_reference above) -- on random, smooth, few-colour and single-colour images, every depth."""
    from PIL import Image
    from dither_pie_amd.dithering_lib import ColorReducer
    from oracle import oracle as orc
    assert ColorReducer._pyset_replay_ok(), "the replay has to match this interpreter's sets (CPython 3.8 ... 3.12)"
    rs = np.random.RandomState(4)
    images = [orc.rnd(90, 120, 3), orc.imgl(120, 160, 5, "smooth"), orc.grad(64, 96), rs.randint(0, 3, (40, 40, 3)).astype(np.uint8),
              np.full((8, 8, 3), 200, np.uint8), orc.rnd(300, 300, 8)]
    try:
        for img in images:
            for n in (1, 2, 3, 16, 20, 256, 1024):
                ColorReducer._replay_ok = True
                a = ColorReducer.reduce_colors(Image.fromarray(img), n)
                ColorReducer._replay_ok = False
                b = ColorReducer.reduce_colors(Image.fromarray(img), n)
                assert a == b, (img.shape, n)
                assert all(isinstance(v, int) for c in a for v in c)
    finally:
        ColorReducer._replay_ok = None


def test_pyset_order_replay(tmp_path):
    """dp_pyset_order_host against this interpreter's own sets: sizes around every growth step of CPython's table (8 slots,
    x4 growth below 50 000 entries, x2 above), with and without duplicates."""
    from dither_pie_amd import _lib
    L = _lib.load()
    rs = np.random.RandomState(11)
    for n, top in [(1, 256), (4, 256), (5, 256), (6, 4), (19, 256), (20, 256), (77, 256), (308, 256), (1229, 256), (4916, 256),
                   (19661, 256), (52000, 256), (78644, 256), (150000, 256), (30000, 5)]:
        arr = np.ascontiguousarray(rs.randint(0, top, (n, 3)).astype(np.uint8))
        order = np.empty(n, np.uint32)
        nd = C.c_int64(0)
        assert L.dp_pyset_order_host(arr.ctypes.data, n, order.ctypes.data, C.byref(nd)) == 0
        real = list(set(zip(arr[:, 0].tolist(), arr[:, 1].tolist(), arr[:, 2].tolist())))
        assert nd.value == len(real)
        assert [tuple(int(v) for v in c) for c in arr[order[:nd.value]]] == real, (n, top)


def test_expanded_key_error_bound():
    """The candidate scan of error diffusion for palettes of up to 16 colours ranks by |c|^2 + 2^18 - 2 c.o in three float32
    multiply-adds (ed_nearest.hip.h: ed_key_expanded) and accepts the winner when the second key is more than 0.75 above the
    first.  The claim behind that margin, checked here on the float32 arithmetic itself: for points of the cube the value stays
    inside [2^16, 2^19) (positive: bit patterns order like values) and is within 0.32 of the exact one, tag bits included --
    so a computed gap above 0.75 proves the float64 order."""
    rs = np.random.RandomState(5)

    def fma32(a, b, c):  # float32 fused multiply-add: the product of two float32 values is exact in float64
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)

    worst = 0.0
    for trial in range(24):
        if trial % 2 == 0:
            pal = rs.randint(0, 256, (16, 3)).astype(np.float32)
        else:  # gamma palettes: float32 values anywhere in [0, 255]
            pal = (255.0 * (rs.rand(16, 3) ** 2.2)).astype(np.float32)
        o = np.clip(rs.uniform(-20.0, 275.0, (20000, 3)), 0.0, 255.0).astype(np.float32)  # clamped, as the diffusers' points are
        if trial % 3 == 0:
            o = np.round(o)  # integer points: the tie-rich case
        w = ((pal.astype(np.float64) ** 2).sum(1) + 262144.0).astype(np.float32)
        neg2 = (-2.0 * pal).astype(np.float32)
        for j in range(16):
            key = fma32(neg2[j, 0][None], o[:, 0], fma32(neg2[j, 1][None], o[:, 1], fma32(neg2[j, 2][None], o[:, 2], w[j][None])))
            assert key.min() >= 65536.0 and key.max() < 524288.0
            tagged = ((key.view(np.uint32) & ~np.uint32(7)) | np.uint32(7)).view(np.float32)  # the largest tag
            exact = ((pal[j].astype(np.float64)[None] - o.astype(np.float64)) ** 2).sum(1) - (o.astype(np.float64) ** 2).sum(1) + 262144.0
            worst = max(worst, float(np.abs(key.astype(np.float64) - exact).max()), float(np.abs(tagged.astype(np.float64) - exact).max()))
    assert worst < 0.32, worst


def test_pil_image_from_the_staging_buffer_owns_its_pixels():
    """apply_dithering reuses its staging buffers from call to call: the image it hands out must not be a view of one.
    The packed fallback path: Image.fromarray(buffer) -- packed bytes are always unpacked into storage of the image's own.
    The RGBX path (round 5): Image.frombuffer('RGBX', ..., 'raw', 'RGBX', 0, 1) IS a view (that is why it costs nothing) and
    convert('RGB') gives an image that owns its pixels; and _pil_rgbx_into writes exactly Pillow's four-bytes-per-pixel rows (R, G, B, pad) without ever
    building a bytes object of the whole image."""
    from PIL import Image
    from dither_pie_amd.dithering_lib import _pil_rgbx_into
    buf = np.random.RandomState(3).randint(0, 256, 37 * 53 * 3, dtype=np.uint8)
    want = buf.copy().reshape(37, 53, 3)
    img = Image.fromarray(buf.reshape(37, 53, 3), "RGB")
    buf[:] = 0
    assert np.array_equal(np.asarray(img), want)
    # in: a 53 x 37 image and one whose rows are longer than one encoder piece
    for hh, ww in ((37, 53), (3, 400_000)):
        a = np.random.RandomState(hh).randint(0, 256, (hh, ww, 3), dtype=np.uint8)
        host = np.full(hh * ww * 4, 7, np.uint8)
        assert _pil_rgbx_into(Image.fromarray(a, "RGB"), host)
        assert np.array_equal(host.reshape(hh, ww, 4)[..., :3], a)
        assert not _pil_rgbx_into(Image.fromarray(a, "RGB"), np.empty(hh * ww * 4 - 4, np.uint8))   # a buffer that is too small
    # out: the mapped image follows the buffer, its copy does not
    out4 = np.empty((37, 53, 4), np.uint8)
    out4[..., :3] = want
    out4[..., 3] = 255
    view = Image.frombuffer("RGBX", (53, 37), out4.reshape(-1), "raw", "RGBX", 0, 1)
    own = view.convert("RGB")
    out4[..., :3] = 0
    assert own.mode == "RGB" and np.array_equal(np.asarray(own), want)
    assert not np.array_equal(np.asarray(view)[..., :3], want)   # (a view: what apply_dithering must not return)
