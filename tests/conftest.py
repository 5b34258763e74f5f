import json
import os
import sys

import numpy as np
import pytest

# The suite runs on the library that ships, libditherpie_hip.so.  Only the tests that force a table / kernel / schedule
# through a DP_* switch take the `switches` fixture below, which maps the -DDP_EXPERIMENTS twin (the only build that reads
# those variables) for the duration of that one test.
os.environ.pop("DITHER_PIE_EXPERIMENTS", None)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _drop_device_objects():
    from dither_pie_amd import dithering_lib
    dithering_lib.drop_device_caches()


@pytest.fixture
def switches(monkeypatch):
    """monkeypatch, with load() returning libditherpie_hip_exp.so while the test runs (DP_* variables mean something
    there only).  Cached palettes / thresholds are dropped on both sides of the switch: a device object stays with the
    library that made it."""
    from dither_pie_amd import _lib
    _drop_device_objects()
    prev = _lib.select(True)
    assert _lib.load() is not None and _lib.LIB_PATH.endswith("libditherpie_hip_exp.so")
    try:
        yield monkeypatch
    finally:
        _drop_device_objects()
        _lib.select(prev)


PRODUCT_LIBRARY_TESTS = []   # node ids of the GPU tests that ran with the product library mapped (reported at the end)


@pytest.fixture(autouse=True)
def _which_library(request):
    """Every GPU test that does not ask for `switches` must find the product library selected."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from dither_pie_amd import _lib
    if "switches" in request.fixturenames:
        yield
        return
    assert not _lib.EXPERIMENTS and _lib.LIB_PATH.endswith("libditherpie_hip.so"), _lib.LIB_PATH
    yield
    assert not _lib.EXPERIMENTS and _lib.LIB_PATH.endswith("libditherpie_hip.so"), "a test left the twin library selected"
    PRODUCT_LIBRARY_TESTS.append(request.node.nodeid)


def pytest_terminal_summary(terminalreporter):
    if PRODUCT_LIBRARY_TESTS:
        terminalreporter.write_line(f"{len(PRODUCT_LIBRARY_TESTS)} GPU tests ran on libditherpie_hip.so (the product library); "
                                    "the rest took the `switches` fixture (libditherpie_hip_exp.so)")


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gold():
    return np.load(os.path.join(GOLDEN, "small.npz"))


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


def case_input(orc, spec):
    if spec[0] == "rnd":
        return orc.rnd(spec[1], spec[2], spec[3])
    if spec[0] == "grad":
        return orc.grad(spec[1], spec[2])
    if spec[0] == "imgl":
        return orc.imgl(spec[1], spec[2], spec[3], spec[4])
    raise ValueError(spec)


def case_palette(orc, spec):
    if spec[0] == "U":
        return orc.generate_uniform_palette(spec[1])
    if spec[0] == "palr":
        return orc.palr(spec[1], spec[2] if len(spec) > 2 else 7)
    if spec[0] == "list":
        return [tuple(c) for c in spec[1]]
    raise ValueError(spec)


def numba_fixtures():
    """(cases dict, arrays) recorded by `tests/golden/make_golden.py --numba` on a host with numba, or None: the build container
    has no numba and cannot install it, so until somebody runs that one command the numba branches stay parity-unpinned."""
    import json
    jp, zp = os.path.join(GOLDEN, "numba.json"), os.path.join(GOLDEN, "numba.npz")
    if not (os.path.exists(jp) and os.path.exists(zp)):
        return None
    with open(jp) as f:
        return json.load(f), np.load(zp)


def fake_ffmpeg_tools(tmp_path, monkeypatch, frames, rotation=None, coded=None, encoder_dies_after=None, trailing=b"", encoder_sleeps=0.0):
    """Stand-ins for ffmpeg / ffprobe on PATH (there is no ffmpeg in the image): the decoder emits `frames` as rgb24, the
    encoder stores `-s` and the bytes it receives, ffprobe answers the queries get_video_info / _probe_rotation make.
    encoder_dies_after: the encoder exits with code 3 after that many bytes (a crashed ffmpeg); trailing: bytes the
    decoder appends to the stream (a stream that is not a whole number of frames); encoder_sleeps: seconds the encoder waits
    before it reads its first byte and again after every 100 000 bytes (a codec that cannot keep up)."""
    import stat
    import sys
    n, h, w = frames.shape[:3]
    cw, ch = coded if coded else (w, h)
    raw = tmp_path / "input.raw"
    raw.write_bytes(frames.tobytes() + trailing)
    fake_ffmpeg = tmp_path / "ffmpeg"
    fake_ffmpeg.write_text(f"""#!{sys.executable}
import sys
a = sys.argv[1:]
if "pipe:1" in a:      # decoder: raw frames to stdout; it must have been asked for the displayed geometry
    assert a[a.index("-s") + 1] == "{w}x{h}", a
    assert "-noautorotate" not in a
    sys.stdout.buffer.write(open({str(raw)!r}, "rb").read())
elif "pipe:0" in a:    # encoder: keep the size argument and the bytes
    if {encoder_dies_after!r} is not None:
        sys.stdin.buffer.read({encoder_dies_after!r})
        sys.exit(3)
    if {encoder_sleeps!r}:
        import time
        got = []
        while True:
            time.sleep({encoder_sleeps!r})
            b = sys.stdin.buffer.read(100000)
            if not b:
                break
            got.append(b)
        open(a[-1], "wb").write(a[a.index("-s") + 1].encode() + b"\\n" + b"".join(got))
    else:
        open(a[-1], "wb").write(a[a.index("-s") + 1].encode() + b"\\n" + sys.stdin.buffer.read())
else:
    sys.exit(2)
""")
    fake_ffprobe = tmp_path / "ffprobe"
    fake_ffprobe.write_text(f"""#!{sys.executable}
import sys
e = sys.argv[sys.argv.index("-show_entries") + 1]
print({{"stream=r_frame_rate": "25/1", "stream=width,height": "{cw}\\n{ch}", "stream=duration,nb_frames": "0.44\\n{n}",
       "stream_tags=rotate:stream_side_data=rotation": {("" if rotation is None else str(rotation))!r}}}[e])
""")
    for f in (fake_ffmpeg, fake_ffprobe):
        f.chmod(f.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv("PATH", str(tmp_path) + ":" + os.environ["PATH"])
