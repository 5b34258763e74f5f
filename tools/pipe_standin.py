"""Build and drive tools/pipe_standin.c: compiled stand-ins for ffmpeg / ffprobe on the rawvideo-pipe path.
Test and bench infrastructure (there is no ffmpeg in the image); never imported by dither_pie_amd."""
from __future__ import annotations

import os
import shutil
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCE = os.path.join(HERE, "pipe_standin.c")


def build(directory: str) -> str:
    """gcc -O3 the stand-in into `directory` as `ffmpeg` and `ffprobe`; returns the directory (to be put on PATH)."""
    os.makedirs(directory, exist_ok=True)
    exe = os.path.join(directory, "ffmpeg")
    subprocess.check_call(["gcc", "-O3", "-march=native", "-o", exe, SOURCE])
    shutil.copy2(exe, os.path.join(directory, "ffprobe"))
    return directory


def environment(directory: str, n_frames: int, h: int, w: int, distinct: int = 8, keep: int = 2, base=None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update({"PATH": directory + ":" + env.get("PATH", ""), "DP_STANDIN_W": str(w), "DP_STANDIN_H": str(h),
                "DP_STANDIN_FRAMES": str(n_frames), "DP_STANDIN_DISTINCT": str(distinct), "DP_STANDIN_KEEP": str(keep)})
    return env


def frames(n_frames: int, h: int, w: int, distinct: int = 8, only=None) -> np.ndarray:
    """The decoder stand-in's stream as [n, h, w, 3] uint8 (pipe_standin.c: decoder); `only`: just these frame numbers."""
    fb = h * w * 3
    k = max(1, min(distinct, n_frames))
    which = list(range(n_frames)) if only is None else list(only)
    out = np.empty((len(which), fb), np.uint8)
    i = np.arange(fb, dtype=np.uint64)
    for j, f in enumerate(which):
        v = ((i + np.uint64((f % k) * fb)) & np.uint64(0xffffffff)) * np.uint64(2654435761) & np.uint64(0xffffffff)
        out[j] = (v >> np.uint64(24)).astype(np.uint8)
        out[j, :4] = np.frombuffer(np.uint32(f).tobytes(), np.uint8)
    return out.reshape(len(which), h, w, 3)


def weighted_sum(frames_u8) -> int:
    """The encoder stand-in's order-sensitive checksum of a [n, ...] uint8 stream: sum((f + 1) * bytesum(frame f)) mod 2^64."""
    total = 0
    for f, fr in enumerate(frames_u8):
        total = (total + (f + 1) * int(np.asarray(fr, dtype=np.uint8).sum(dtype=np.uint64))) & 0xffffffffffffffff
    return total


def read_summary(path: str):
    """-> (dict of the encoder stand-in's summary line, kept frames [k, H, W, 3])"""
    with open(path, "rb") as f:
        head, body = f.read().split(b"\n", 1)
    parts = head.decode().split()
    w, h = (int(v) for v in parts[0].split("x"))
    info = {"w": w, "h": h}
    for p in parts[1:]:
        k, v = p.split("=")
        info[k] = int(v)
    kept = np.frombuffer(body, np.uint8).reshape(info["keep"], h, w, 3)
    return info, kept
