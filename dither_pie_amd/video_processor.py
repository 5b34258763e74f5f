"""Drop-in counterpart of dither_pie's `video_processor` for the frame path, on MI355X.

Same public names and call signatures as the reference module (video_processor.py:27-46, 98, 172-178,
393-475, 547-577).  What changes is where frames are processed: the reference forks a
multiprocessing.Pool of <= 4 workers per 15-frame batch and pickles the ditherer into each
(video_processor.py:304-346); a HIP context must not cross fork(), and frames are independent, so here
frames are processed IN PROCESS in batches that stay resident in HBM between the stages
(NEAREST down-scale -> dither -> NEAREST up-scale, all through libditherpie_hip.so).
ffmpeg/ffprobe remain external subprocesses as in the reference (same libx264 re-encode with audio/subtitle
copy), but frames travel as rawvideo rgb24 through pipes into pinned host buffers instead of PNG files on
disk (PNG encode/decode dominates a processed frame in the reference); the PNG-file
exchange of the reference remains available (`use_pipes=False`).  Both keep the reference's failure policy
(video_processor.py:325-336, 53-96): a batch that fails is retried frame by frame (three attempts each), a frame that
still fails is replaced by the nearest good output frame (the previous one first) and the video goes on; the call
returns False only when ffmpeg fails or no frame could be processed.  progress_callback(fraction, message) gets the
same milestones (0.0, 0.05, 0.1 ... 0.9, 1.0).
"""
from __future__ import annotations

import shutil
import subprocess
import sys
import tempfile
from multiprocessing import cpu_count
from pathlib import Path
from typing import Callable, Optional, Tuple

import numpy as np
from PIL import Image

from .dithering_lib import ImageDitherer, PixelizeMethod

__all__ = ["VideoProcessor", "pixelize_regular", "NeuralPixelizer", "process_frames"]


def _even_dimensions(orig_w: int, orig_h: int, max_size: int) -> Tuple[int, int]:
    """video_processor.py:547-560: the smaller side becomes max_size (made even), the other keeps the
    aspect ratio, rounded and made even."""
    small = max_size if max_size % 2 == 0 else max_size - 1
    if orig_w >= orig_h:
        other = int(round((orig_w / orig_h) * small))
        return other + (other % 2), small
    other = int(round((orig_h / orig_w) * small))
    return small, other + (other % 2)


class NeuralPixelizer:
    """The GAN pixelizer (video_processor.py:478-545) is a separate model-inference workload whose
    weights are not part of the repository; only the dimension helper is provided."""

    _compute_even_dimensions = staticmethod(_even_dimensions)

    def __init__(self, *a, **k):
        raise NotImplementedError("neural pixelization is outside the MI355X backend's scope")


def pixelize_regular(image: Image.Image, max_size: int) -> Image.Image:
    """NEAREST down-scale to even dimensions (video_processor.py:563-577), on the GPU."""
    import torch
    from . import backend
    tw, th = _even_dimensions(image.size[0], image.size[1], max_size)
    arr = np.array(image.convert("RGB"), dtype=np.uint8)
    out = backend.resize_nearest(torch.from_numpy(arr).cuda(), th, tw)
    return Image.fromarray(out.cpu().numpy(), "RGB")


def _final_size(w: int, h: int, multiplier: int) -> Tuple[int, int]:
    """video_processor.py:408-415: integer multiple, bumped to even for yuv420p."""
    nw, nh = w * multiplier, h * multiplier
    return nw + (nw % 2), nh + (nh % 2)


def output_size(h: int, w: int, pixelize_method: Optional[str] = None, max_size: int = 64,
                final_resize_multiplier: Optional[int] = None) -> Tuple[int, int]:
    """(H', W') of process_frames() for frames of h x w: the geometry of video_processor.py:423-475 without running it."""
    if pixelize_method in (PixelizeMethod.REGULAR.value, "regular"):
        w, h = _even_dimensions(w, h, max_size)
    if final_resize_multiplier:
        w, h = _final_size(w, h, final_resize_multiplier)
    return h, w


def _apply_final_resize_to_frame(image: Image.Image, multiplier: int) -> Image.Image:
    """video_processor.py:393-420"""
    import torch
    from . import backend
    nw, nh = _final_size(image.size[0], image.size[1], multiplier)
    arr = np.array(image.convert("RGB"), dtype=np.uint8)
    out = backend.resize_nearest(torch.from_numpy(arr).cuda(), nh, nw)
    return Image.fromarray(out.cpu().numpy(), "RGB")


def process_frames(frames, ditherer: ImageDitherer, pixelize_method: Optional[str] = None, max_size: int = 64,
                   final_resize_multiplier: Optional[int] = None):
    """The per-frame pipeline of _process_single_frame (video_processor.py:423-475) for a batch of
    equally sized frames resident in HBM: uint8 CUDA tensor [N,H,W,3] -> uint8 CUDA tensor [N,H',W',3]."""
    from . import backend
    if pixelize_method in (PixelizeMethod.NEURAL.value, "neural"):
        raise NotImplementedError("neural pixelization is outside the MI355X backend's scope")
    x = frames
    if pixelize_method in (PixelizeMethod.REGULAR.value, "regular"):
        tw, th = _even_dimensions(x.shape[2], x.shape[1], max_size)
        x = backend.resize_nearest(x, th, tw)
    x = ditherer.apply_dithering_frames(x)
    if final_resize_multiplier:
        nw, nh = _final_size(x.shape[2], x.shape[1], final_resize_multiplier)
        x = backend.resize_nearest(x, nh, nw)
    return x


def _process_single_frame(frame_path: Path, ditherer: ImageDitherer, pixelize_method: Optional[str] = None,
                          max_size: int = 64, final_resize_multiplier: Optional[int] = None) -> bool:
    """One PNG in place; True on success, False (with a message on stderr) on any error
    (video_processor.py:423-475)."""
    try:
        import torch
        frame_path = Path(frame_path)
        arr = np.array(Image.open(frame_path).convert("RGB"), dtype=np.uint8)
        out = process_frames(torch.from_numpy(arr).cuda().unsqueeze(0), ditherer, pixelize_method, max_size,
                             final_resize_multiplier)
        Image.fromarray(out[0].cpu().numpy(), "RGB").save(frame_path)
        if not frame_path.exists() or frame_path.stat().st_size == 0:
            raise ValueError(f"Frame {frame_path} not saved properly")
        return True
    except Exception as e:  # noqa: BLE001 - the reference swallows everything here too
        print(f"Error processing frame {frame_path}: {e}", file=sys.stderr)
        return False


class _PipeSplicer:
    """vmsplice(2) of a buffer's pages into a pipe: the kernel takes references to the pages instead of copying them into pipe
    buffers of its own -- no copy and no page allocation on the writing side, which on the bench box also speeds the READER of
    the other pipe up by a third (the two kernel-side copies contend; tools/bench_scripts/pipe_vmsplice.py: writer 650 -> 1890
    fps, concurrent reader 620 -> 800).  The pages stay referenced until the pipe's reader has consumed them, so the CALLER must
    not overwrite the buffer before that (the overlapped pipe path holds a slot back until consumed_bytes() has passed it).
    Linux only; anything unexpected turns it off and the caller writes as before."""

    def __init__(self):
        self.ok = False
        try:
            import ctypes
            import ctypes.util
            libc = ctypes.CDLL(ctypes.util.find_library("c") or "libc.so.6", use_errno=True)

            class IoVec(ctypes.Structure):
                _fields_ = [("base", ctypes.c_void_p), ("len", ctypes.c_size_t)]

            libc.vmsplice.argtypes = [ctypes.c_int, ctypes.POINTER(IoVec), ctypes.c_ulong, ctypes.c_uint]
            libc.vmsplice.restype = ctypes.c_ssize_t
            self._ct, self._libc, self._IoVec = ctypes, libc, IoVec
            self.ok = sys.platform.startswith("linux")
        except Exception:  # noqa: BLE001
            pass

    def splice_all(self, fd: int, addr: int, nbytes: int) -> bool:
        """All nbytes at addr into the pipe fd.  True: done.  False: not available here (nothing was written: the caller
        falls back to write()).  Raises BrokenPipeError when the reader is gone."""
        import errno
        ct = self._ct
        off = 0
        while off < nbytes:
            iov = self._IoVec(addr + off, nbytes - off)
            n = self._libc.vmsplice(fd, ct.byref(iov), 1, 0)
            if n < 0:
                e = ct.get_errno()
                if e == errno.EINTR:
                    continue
                if e == errno.EPIPE:
                    raise BrokenPipeError(e, "vmsplice: the encoder closed its pipe")
                if off == 0:
                    self.ok = False   # EINVAL / EFAULT / ENOSYS: not here
                    return False
                raise OSError(e, "vmsplice failed in the middle of a batch")
            off += n
        return True

    @staticmethod
    def unread_bytes(fd: int) -> int:
        import array
        import fcntl
        import termios
        buf = array.array("i", [0])
        fcntl.ioctl(fd, termios.FIONREAD, buf)
        return int(buf[0])


class VideoProcessor:
    """video_processor.py:27-390"""

    def __init__(self, num_workers: Optional[int] = None,
                 progress_callback: Optional[Callable[[float, str], None]] = None, devices=None):
        if num_workers is None:
            num_workers = min(4, max(1, cpu_count() - 1))
        self.num_workers = num_workers  # kept for interface compatibility; frames batch on the GPU instead
        self.progress_callback = progress_callback
        # an addition: the GPUs the frame batches are spread over (contiguous blocks, one worker thread and stream per
        # device -- the reference's Pool over frames, video_processor.py:304-346).  None = every visible device.
        self.devices = devices

    def _devices(self):
        from . import sharding
        return sharding.visible_devices() if self.devices is None else list(self.devices)

    def _report_progress(self, fraction: float, message: str):
        if self.progress_callback:
            self.progress_callback(fraction, message)

    def _fix_failed_frames(self, failed_frames: list, all_frames: list):
        """Copy the nearest good frame (previous first, then next) over each failed one
        (video_processor.py:53-96)."""
        bad = set(failed_frames)
        for f in failed_frames:
            if f not in all_frames:
                print(f"Could not find index for {f.name}", file=sys.stderr)
                continue
            i = all_frames.index(f)
            order = list(range(i - 1, -1, -1)) + list(range(i + 1, len(all_frames)))
            src = next((all_frames[j] for j in order if all_frames[j] not in bad and all_frames[j].exists()), None)
            if src is None:
                print(f"ERROR: Could not find any successful frame to copy for {f.name}", file=sys.stderr)
                continue
            try:
                shutil.copy2(src, f)
                print(f"Fixed {f.name} by copying from {src.name}", file=sys.stderr)
            except Exception as e:  # noqa: BLE001
                print(f"Failed to copy frame {src.name} to {f.name}: {e}", file=sys.stderr)

    def get_video_info(self, video_path: str) -> dict:
        """fps / width / height / duration / frame_count via ffprobe, with the reference's defaults when
        probing fails (video_processor.py:98-170)."""
        def probe(entries):
            r = subprocess.run(["ffprobe", "-v", "error", "-select_streams", "v:0", "-show_entries",
                                f"stream={entries}", "-of", "default=nokey=1:noprint_wrappers=1", video_path],
                               capture_output=True, text=True, check=True)
            return r.stdout.strip()
        try:
            rate = probe("r_frame_rate")
            if "/" in rate:
                a, b = rate.split("/")
                fps = float(a) / float(b)
            else:
                fps = float(rate) if rate else 30.0
            dims = probe("width,height").split("\n")
            width = int(dims[0]) if len(dims) > 0 else 1920
            height = int(dims[1]) if len(dims) > 1 else 1080
            duration = frame_count = None
            for line in probe("duration,nb_frames").split("\n"):
                if line and line != "N/A":
                    try:
                        v = float(line)
                    except ValueError:
                        continue
                    if v > 100:
                        frame_count = int(v)
                    else:
                        duration = v
            if frame_count is None and duration is not None:
                frame_count = int(duration * fps)
            self._probe_ok = True
            return {"fps": fps, "width": width, "height": height, "duration": duration, "frame_count": frame_count}
        except Exception as e:  # noqa: BLE001
            print(f"Warning: Could not get video info: {e}", file=sys.stderr)
            # the reference's defaults (video_processor.py:164-170); _probe_ok tells the pipe path that width and
            # height are guesses it must not slice a byte stream with
            self._probe_ok = False
            return {"fps": 30.0, "width": 1920, "height": 1080, "duration": None, "frame_count": None}

    def _process_batch(self, files, ditherer, pixelize_method, max_size, final_resize_multiplier):
        """One batch of PNG files through the GPU; returns the list of files that failed."""
        import torch
        try:
            arrs = [np.array(Image.open(f).convert("RGB"), dtype=np.uint8) for f in files]
            if len({a.shape for a in arrs}) != 1:
                raise ValueError("frames of one batch differ in size")
            out = process_frames(torch.from_numpy(np.stack(arrs)).cuda(), ditherer, pixelize_method, max_size,
                                 final_resize_multiplier).cpu().numpy()
            for f, o in zip(files, out):
                Image.fromarray(o, "RGB").save(f)
                if not f.exists() or f.stat().st_size == 0:
                    raise ValueError(f"Frame {f} not saved properly")
            return []
        except Exception as e:  # noqa: BLE001
            print(f"Batch failed ({e}); retrying frame by frame", file=sys.stderr)
        failed = []
        for f in files:
            if not any(_process_single_frame(f, ditherer, pixelize_method, max_size, final_resize_multiplier)
                       for _ in range(3)):
                failed.append(f)
        return failed

    ATTEMPTS = 3  # the first try plus the reference's two retries (video_processor.py:325-336)

    @staticmethod
    def _device_is_gone(e: BaseException) -> bool:
        """A failure of the DEVICE, not of a frame: the library's DP_EHIP / DP_ENOMEM, or a HIP / out-of-memory error out of
        torch.  The reference's per-frame retry policy is about PNG and I/O failures; retrying on a faulted or exhausted GPU
        only makes 3 x batch doomed launches and would fill the rest of the video with copies of the last good frame."""
        from ._lib import DP_EHIP, DP_ENOMEM, DitherPieError
        if isinstance(e, DitherPieError):
            return e.code in (DP_EHIP, DP_ENOMEM)
        try:
            import torch
            if isinstance(e, torch.cuda.OutOfMemoryError):
                return True
        except Exception:  # noqa: BLE001
            pass
        text = str(e)
        return isinstance(e, RuntimeError) and any(k in text for k in ("HIP error", "CUDA error", "hipError", "out of memory"))

    def _batch_with_retries(self, host_frames, run, to_host=None):
        """The reference's failure policy for one batch (video_processor.py:304-346): the whole batch in one go; if that
        raises, every frame on its own, up to ATTEMPTS times; a frame that keeps failing is reported as None and the
        caller substitutes a neighbour.  run(frames [k,H,W,3]) -> [k,H',W',3]; to_host(frame tensor) -> host tensor (the
        default is .cpu() on the current stream; a caller whose run() works on a stream of its own passes its own).
        -> (out tensor [n,...] or None, per-frame list or None): exactly one of the two is set."""
        if to_host is None:
            def to_host(t):
                return t.cpu()
        try:
            return run(host_frames), None
        except Exception as e:  # noqa: BLE001 - "a frame that errors must not abort the video"
            if self._device_is_gone(e):
                raise   # not a frame's fault: process_video_streaming reports failure instead of substituting frames
            print(f"Batch failed ({e}); retrying frame by frame", file=sys.stderr)
        outs = []
        for i in range(host_frames.shape[0]):
            o = None
            for attempt in range(self.ATTEMPTS):
                try:
                    o = to_host(run(host_frames[i:i + 1])[0])
                    break
                except Exception as e:  # noqa: BLE001
                    if self._device_is_gone(e):
                        raise
                    print(f"Error processing frame {i} of the batch (attempt {attempt + 1}): {e}", file=sys.stderr)
            outs.append(o)
        return None, outs

    def _probe_rotation(self, video_path: str) -> int:
        """Display rotation of the first video stream in degrees (0 when there is none): ffmpeg auto-rotates while
        decoding - the reference's plain `ffmpeg -i input frame_%05d.png` (video_processor.py:208-217) does - so the
        piped frames have the ROTATED geometry."""
        try:
            r = subprocess.run(["ffprobe", "-v", "error", "-select_streams", "v:0", "-show_entries",
                                "stream_tags=rotate:stream_side_data=rotation", "-of",
                                "default=nokey=1:noprint_wrappers=1", video_path], capture_output=True, text=True, check=True)
            for line in r.stdout.split("\n"):
                try:
                    return int(round(float(line.strip()))) % 360
                except ValueError:
                    continue
        except Exception as e:  # noqa: BLE001
            print(f"Warning: Could not probe rotation: {e}", file=sys.stderr)
        return 0

    PIPE_SLOTS = 4            # rotating pinned batch slots of the overlapped pipe path (reader, GPU, writer; the writer keeps a
                              # spliced slot until the encoder has consumed it)
    PIPE_ZERO_COPY = True     # vmsplice the output slots into the encoder pipe (see _PipeSplicer)
    PIPE_SLOT_BYTES = 128 << 20   # pinned bytes of a slot's input (or output) buffer at most
    PIPE_BYTES = 1 << 20      # requested pipe capacity (F_SETPIPE_SZ; the kernel default is 64 KiB = one syscall per 64 KiB)

    @staticmethod
    def _widen_pipe(fileobj):
        """Best effort: a 1 MiB pipe instead of 64 KiB, so that a 6 MB frame is ~6 reads / writes instead of ~95."""
        try:
            import fcntl
            fcntl.fcntl(fileobj.fileno(), getattr(fcntl, "F_SETPIPE_SZ", 1031), VideoProcessor.PIPE_BYTES)
        except Exception:  # noqa: BLE001 - not Linux, or above /proc/sys/fs/pipe-max-size
            pass

    @staticmethod
    def _write_all(f, buf):
        """The whole buffer into an unbuffered pipe (a raw write may be partial when a signal arrives)."""
        mv = memoryview(buf).cast("B")
        while len(mv):
            n = f.write(mv)
            mv = mv[len(mv) if n is None else n:]

    def _stream_through_pipes(self, input_path, output_path, ditherer, method, max_size, batch_size,
                              final_resize_multiplier, info, run=None, overlap=True) -> int:
        """decode -> GPU -> encode through two ffmpeg rawvideo pipes; returns the number of frames written.
        Failure policy as in the reference: a batch that fails is retried frame by frame, a frame that still fails is
        replaced by the nearest good output frame (the previous one first, video_processor.py:53-96) and the video goes
        on; the call fails only when ffmpeg does or when no frame at all could be processed.  `run` (tests): the batch
        function, default process_frames on the configured devices.

        overlap=True (the default): the three stages run CONCURRENTLY on PIPE_SLOTS rotating pinned batch slots -- a reader
        thread fills a slot from the decoder pipe, this thread submits it to the GPU on a stream of its own (H2D,
        process_frames, D2H into the slot's pinned output buffer, one event; it never waits for the device), a writer
        thread waits for the event, applies the substitution policy in frame order and feeds the encoder pipe -- so a
        video costs max(decode, GPU, encode) per batch instead of their sum (the reference overlaps nothing either: it
        extracts every PNG, then processes, then encodes, video_processor.py:204-217, 304-346, 361-382).
        overlap=False: the serial loop of rounds 2-4 (read -> H2D -> kernels -> D2H -> write per batch), kept as the
        byte-for-byte A/B partner of the overlapped path.  Both fill self.last_pipe_stats."""
        import time
        import torch
        w, h, fps = int(info["width"]), int(info["height"]), info["fps"]
        if self._probe_rotation(input_path) in (90, 270):
            w, h = h, w  # ffprobe reports the coded size; the decoder below rotates as the reference's extraction does
        frame_bytes = w * h * 3
        total_hint = info.get("frame_count") or 0
        devs = self._devices() if run is None else [None]
        batch_size = batch_size * len(devs)  # one batch per device in flight
        product_run = run is None
        pin = torch.cuda.is_available()
        out_geom = None   # (H', W') of the product path: known up front, so that the output slots are allocated once
        if product_run:
            # a slot is one batch of input + one of output, pinned; three of them rotate: keep a slot's larger half under
            # PIPE_SLOT_BYTES (4K frames: 5 per batch instead of the reference's 15 -- only the granularity of the retry
            # policy changes with it, not the result)
            oh_, ow_ = output_size(h, w, method, max_size, final_resize_multiplier)
            per_frame = max(frame_bytes, oh_ * ow_ * 3)
            batch_size = max(len(devs), min(batch_size, self.PIPE_SLOT_BYTES // max(per_frame, 1)))
        gpu_stream = None  # overlap, one device: the stream this call's H2D / kernels / D2H are queued on
        target = {"out": None}   # several devices: the pinned host tensor the workers write the current batch into
        if product_run:
            out_geom = output_size(h, w, method, max_size, final_resize_multiplier)
            if overlap and len(devs) == 1 and pin:
                with torch.cuda.device(devs[0]):
                    gpu_stream = torch.cuda.Stream()  # the caller's current stream is never blocked or waited on

            def run(x):  # noqa: F811 - the product batch function: frames stay in HBM between the stages
                if len(devs) > 1 and x.shape[0] > 1:
                    from . import sharding
                    out = target["out"]
                    return sharding.process_on_devices(
                        x, lambda y: process_frames(y, ditherer, method, max_size, final_resize_multiplier), devs,
                        out=None if out is None else out[:x.shape[0]])
                with torch.cuda.device(devs[0]):
                    if gpu_stream is not None:
                        with torch.cuda.stream(gpu_stream):
                            return process_frames(x.cuda(non_blocking=True), ditherer, method, max_size, final_resize_multiplier)
                    return process_frames(x.cuda(non_blocking=True), ditherer, method, max_size, final_resize_multiplier)
        def to_host(t):
            if gpu_stream is not None and t.is_cuda:   # the copy goes behind the kernels, on their stream
                with torch.cuda.device(t.device), torch.cuda.stream(gpu_stream):
                    return t.cpu()
            return t.cpu()

        # the byte stream is sliced into frames of exactly w x h: the decoder's output size is made explicit
        dec = subprocess.Popen(["ffmpeg", "-v", "error", "-i", input_path, "-f", "rawvideo", "-pix_fmt",
                                "rgb24", "-s", f"{w}x{h}", "pipe:1"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                               bufsize=0)
        self._widen_pipe(dec.stdout)
        state = {"enc": None, "last_good": None, "leading": 0, "substituted": 0, "written": 0, "piped": 0}
        stats = {"mode": "overlapped" if overlap else "serial", "slots": self.PIPE_SLOTS if overlap else 1, "batch_frames": batch_size,
                 "read_s": 0.0, "gpu_submit_s": 0.0, "gpu_wait_s": 0.0, "write_s": 0.0, "frames": 0}
        self.last_pipe_stats = stats

        def open_encoder(shape):
            oh, ow = int(shape[0]), int(shape[1])
            enc = subprocess.Popen(
                ["ffmpeg", "-y", "-v", "error", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", f"{ow}x{oh}",
                 "-framerate", f"{fps:.5f}", "-i", "pipe:0", "-i", input_path, "-map", "0:v:0", "-map", "1:a?",
                 "-map", "1:s?", "-c:v", "libx264", "-preset", "medium", "-crf", "18", "-pix_fmt", "yuv420p",
                 "-c:a", "copy", "-c:s", "copy", output_path],
                stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, bufsize=0)
            self._widen_pipe(enc.stdin)
            return enc

        def read_batch(view):
            """-> (whole frames read, bytes read): fills `view` from the decoder until it is full or the stream ends"""
            t0 = time.perf_counter()
            got, want = 0, batch_size * frame_bytes
            while got < want:  # a pipe read returns at most the pipe's capacity at a time
                n = dec.stdout.readinto(view[got:want])
                if not n:
                    break
                got += n
            stats["read_s"] += time.perf_counter() - t0
            if got % frame_bytes:
                raise RuntimeError(f"decoder stream is not a whole number of {w}x{h} rgb24 frames "
                                   f"({got % frame_bytes} bytes left over)")
            return got // frame_bytes, got

        splicer = _PipeSplicer() if (overlap and self.PIPE_ZERO_COPY) else None

        def emit(out_host, n_frames, per_frame, may_splice=False):
            """One batch to the encoder, in frame order, with the substitution policy (state: the encoder, the newest good
            output frame, frames that failed before any frame succeeded).  out_host: host tensor [>= n_frames, H', W', 3]
            of a batch that succeeded as a whole; per_frame: list of host tensors / None of one that was retried.
            may_splice: out_host is a slot's own buffer that the caller keeps untouched until the encoder has consumed it.
            -> True when the batch's pages were handed to the pipe by reference (state["piped"] = bytes piped so far)."""
            t0 = time.perf_counter()
            spliced = False
            if out_host is not None:  # the usual case: one write of the whole batch
                if state["enc"] is None:
                    state["enc"] = open_encoder(out_host.shape[1:])
                for _ in range(state["leading"]):
                    self._write_all(state["enc"].stdin, out_host[0].numpy())
                    state["piped"] += out_host[0].numel()
                state["leading"] = 0
                body = out_host[:n_frames]
                if may_splice and splicer is not None and splicer.ok and body.is_contiguous():
                    spliced = splicer.splice_all(state["enc"].stdin.fileno(), body.data_ptr(), body.numel())
                if not spliced:
                    self._write_all(state["enc"].stdin, body.numpy())
                state["piped"] += body.numel()
                # (numpy's single-threaded copy, not tensor.clone(): a torch CPU op wakes the OpenMP pool -- one thread per
                # visible core, 256 on the bench box -- whose idle spinning burns the container's CPU quota (16 cores) and
                # gets reader, writer and both ffmpeg processes throttled: measured as ~20 ms lost per batch)
                state["last_good"] = torch.from_numpy(out_host[n_frames - 1].numpy().copy())
            else:
                for i, o in enumerate(per_frame):
                    if o is not None:
                        state["last_good"] = o
                    else:
                        state["substituted"] += 1
                        if state["last_good"] is not None:
                            o = state["last_good"]  # the previous good frame first (video_processor.py:66-77)
                        else:  # nothing before it: the next good frame, of this batch or of a later one
                            o = next((q for q in per_frame[i + 1:] if q is not None), None)
                            if o is None:
                                state["leading"] += 1
                                continue
                    if state["enc"] is None:
                        state["enc"] = open_encoder(o.shape)
                    buf = np.ascontiguousarray(o.numpy())
                    for _ in range(state["leading"] + 1):
                        self._write_all(state["enc"].stdin, buf)
                        state["piped"] += buf.size
                    state["leading"] = 0
            state["written"] += n_frames
            stats["write_s"] += time.perf_counter() - t0
            if spliced:
                stats["spliced_batches"] = stats.get("spliced_batches", 0) + 1
            return spliced

        def consumed_bytes():
            """bytes of the encoder pipe's stream that its reader has taken out of the pipe so far"""
            if state["enc"] is None:
                return 0
            return state["piped"] - _PipeSplicer.unread_bytes(state["enc"].stdin.fileno())

        def progress(done):
            frac = done / total_hint if total_hint else 0.5
            self._report_progress(0.1 + 0.8 * min(frac, 1.0), f"Processed {done}/{total_hint or '?'} frames")

        def new_out(shape):
            return torch.empty((batch_size,) + tuple(shape), dtype=torch.uint8, pin_memory=pin)

        done = 0
        t_wall = time.perf_counter()
        try:
            if overlap:
                done = self._pipe_overlapped(batch_size, frame_bytes, (h, w), out_geom, run, to_host, gpu_stream, target, read_batch, emit,
                                             progress, new_out, stats, dec, consumed_bytes, state)
            else:
                stage = torch.empty(batch_size * frame_bytes, dtype=torch.uint8, pin_memory=pin)
                view = memoryview(stage.numpy())
                out_host = new_out(tuple(out_geom) + (3,)) if out_geom is not None else None
                target["out"] = out_host
                while True:
                    n_frames, got = read_batch(view)
                    if n_frames == 0:
                        break
                    host_frames = stage[:n_frames * frame_bytes].view(n_frames, h, w, 3)
                    t0 = time.perf_counter()
                    out, per_frame = self._batch_with_retries(host_frames, run, to_host)
                    if out is not None:  # one copy into the pinned buffer
                        if out_host is None:
                            out_host = new_out(out.shape[1:])
                        if out.is_cuda:
                            out_host[:n_frames].copy_(out, non_blocking=True)
                        elif out.data_ptr() != out_host.data_ptr():   # (a host tensor: numpy's copy, see emit())
                            np.copyto(out_host[:n_frames].numpy(), out.numpy())
                        if out.is_cuda:
                            torch.cuda.current_stream(out.device).synchronize()
                    stats["gpu_wait_s"] += time.perf_counter() - t0
                    emit(out_host if out is not None else None, n_frames, per_frame)   # (one buffer, reused at once: copied, not spliced)
                    done += n_frames
                    progress(done)
                    if got < batch_size * frame_bytes:
                        break
            if state["substituted"]:
                print(f"Fixed {state['substituted']} failed frames by copying from nearest frames", file=sys.stderr)
            self._report_progress(0.9, "Finishing the encode...")
        finally:
            if dec.stdout:
                dec.stdout.close()
            rc_dec = dec.wait()
            rc_enc = 0
            if state["enc"] is not None:
                t0 = time.perf_counter()
                try:
                    state["enc"].stdin.close()
                except BrokenPipeError:
                    pass
                rc_enc = state["enc"].wait()
                stats["encoder_drain_s"] = time.perf_counter() - t0
            state.pop("slots", None)   # (only now may the pinned slot buffers go back to the allocator)
            stats["frames"] = done
            stats["wall_s"] = time.perf_counter() - t_wall
        if rc_dec != 0 or rc_enc != 0:
            raise RuntimeError(f"ffmpeg failed (decoder {rc_dec}, encoder {rc_enc})")
        if done == 0:
            raise ValueError("No frames extracted from video")
        if state["last_good"] is None:
            raise RuntimeError("no frame of the video could be processed")
        return done

    def _pipe_overlapped(self, batch_size, frame_bytes, in_geom, out_geom, run, to_host, gpu_stream, target, read_batch, emit, progress,
                         new_out, stats, dec, consumed_bytes, state) -> int:
        """The three concurrent stages of _stream_through_pipes (see there).  Slots rotate free -> filled -> submitted ->
        free; every blocking queue operation polls an abort flag, so an error in any stage (a dead encoder, a device that
        is gone, a malformed stream) ends the other two instead of leaving them blocked on a queue."""
        import queue
        import threading
        import time
        import torch
        h, w = in_geom
        pin = torch.cuda.is_available()

        class Slot:
            def __init__(self):
                self.inp = torch.empty(batch_size * frame_bytes, dtype=torch.uint8, pin_memory=pin)
                self.view = memoryview(self.inp.numpy())
                self.out = new_out(tuple(out_geom) + (3,)) if out_geom is not None else None

        slots = [Slot() for _ in range(max(2, int(self.PIPE_SLOTS)))]
        # (the last spliced slot's pages may still sit in the encoder pipe when this function returns: the caller keeps the slots
        # -- and with them their pinned memory, out of the allocator's hands -- until the encoder has exited)
        state["slots"] = slots
        free_q, filled_q, write_q = queue.Queue(), queue.Queue(), queue.Queue()
        for s in slots:
            free_q.put(s)
        abort = threading.Event()
        errors = []   # (stage, exception) in the order they happened

        class _Abort(Exception):
            pass

        def take(q):
            while True:
                try:
                    return q.get(timeout=0.05)
                except queue.Empty:
                    if abort.is_set():
                        raise _Abort() from None

        def fail(stage, e):
            errors.append((stage, e))
            abort.set()

        def reader():
            try:
                while True:
                    slot = take(free_q)
                    n_frames, got = read_batch(slot.view)
                    filled_q.put((slot, n_frames, got))
                    if n_frames == 0 or got < batch_size * frame_bytes:
                        return
            except _Abort:
                pass
            except BaseException as e:  # noqa: BLE001 - handed to the submitting thread, which re-raises it
                fail("reader", e)

        def writer():
            held = []   # [(slot, stream offset at which its spliced pages end)]: back to the reader once the encoder is past them
            try:
                while True:
                    item = take(write_q)
                    if item is None:
                        return
                    slot, n_frames, out_host, per_frame, event = item
                    if event is not None:
                        t0 = time.perf_counter()
                        event.synchronize()
                        stats["gpu_wait_s"] += time.perf_counter() - t0
                    own = out_host is not None and slot.out is not None and out_host.data_ptr() == slot.out.data_ptr()
                    if emit(out_host, n_frames, per_frame, may_splice=own):
                        held.append((slot, state["piped"]))
                    else:
                        free_q.put(slot)
                    # a spliced slot's pages are referenced by the pipe until read: with the next batch behind them in a
                    # pipe of at most 1 MiB they always are by now, but it is checked, not assumed (a stalled encoder)
                    if held:
                        seen = consumed_bytes()
                        while held and held[0][1] <= seen:
                            free_q.put(held.pop(0)[0])
                        while len(held) > 1:   # never sit on more than one: wait for the encoder instead of starving the reader
                            if abort.is_set():
                                raise _Abort()
                            time.sleep(0.0005)
                            seen = consumed_bytes()
                            while held and held[0][1] <= seen:
                                free_q.put(held.pop(0)[0])
            except _Abort:
                pass
            except BaseException as e:  # noqa: BLE001
                fail("writer", e)

        t_read = threading.Thread(target=reader, name="dp-pipe-reader", daemon=True)
        t_write = threading.Thread(target=writer, name="dp-pipe-writer", daemon=True)
        t_read.start()
        t_write.start()
        done = 0
        try:
            while True:
                slot, n_frames, got = take(filled_q)
                if n_frames == 0:
                    break
                host_frames = slot.inp[:n_frames * frame_bytes].view(n_frames, h, w, 3)
                t0 = time.perf_counter()
                target["out"] = slot.out
                out, per_frame = self._batch_with_retries(host_frames, run, to_host)
                event, out_host = None, None
                if out is not None and out.is_cuda:
                    # D2H into the slot's pinned buffer behind the kernels, on their stream; one event, no host wait
                    if slot.out is None or tuple(slot.out.shape[1:]) != tuple(out.shape[1:]):
                        slot.out = new_out(out.shape[1:])
                    with torch.cuda.device(out.device):
                        st = gpu_stream if gpu_stream is not None else torch.cuda.current_stream()
                        with torch.cuda.stream(st):
                            slot.out[:n_frames].copy_(out, non_blocking=True)
                            event = torch.cuda.Event()
                            event.record(st)
                    out_host = slot.out
                elif out is not None:
                    out_host = out   # on the host already (several devices: the workers wrote it into slot.out)
                stats["gpu_submit_s"] += time.perf_counter() - t0
                write_q.put((slot, n_frames, out_host, per_frame, event))
                done += n_frames
                progress(done)
                if got < batch_size * frame_bytes:
                    break
            write_q.put(None)
            while t_write.is_alive() and not abort.is_set():
                t_write.join(0.05)
        except _Abort:
            pass
        except BaseException:
            abort.set()
            raise
        finally:
            if abort.is_set():
                try:
                    dec.kill()   # a reader blocked in read() wakes up on the end of the stream
                except Exception:  # noqa: BLE001
                    pass
            t_read.join(5.0)
            t_write.join(5.0)
        if errors:
            raise errors[0][1]
        return done

    def process_video_streaming(self, input_path: str, output_path: str, ditherer: ImageDitherer,
                                pixelize_func=None, batch_size: int = 15,
                                final_resize_multiplier: Optional[int] = None, use_pipes: bool = True) -> bool:
        """video_processor.py:172-390; pixelize_func is the reference's tuple (method_str, max_size) or None.
        use_pipes (an addition): rawvideo pipes instead of the reference's PNG files on disk."""
        if use_pipes:
            info = self.get_video_info(input_path)
            if not getattr(self, "_probe_ok", True):
                use_pipes = False  # frame size unknown: the PNG-file exchange below does not depend on it
        if use_pipes:
            try:
                # the reference's milestones (0.0, 0.05, 0.1 ... 0.9, 1.0; video_processor.py:199-384) with messages that
                # say what this path does at them: decode, processing and encode run concurrently through the pipes
                self._report_progress(0.0, "Initializing video processing...")
                self._report_progress(0.05, "Starting the decoder...")
                method, max_size = (None, 64) if pixelize_func is None else pixelize_func
                self._report_progress(0.1, "Processing frames...")
                self._stream_through_pipes(input_path, output_path, ditherer, method, max_size, max(1, batch_size),
                                           final_resize_multiplier, info)   # reports 0.9 before it waits for the encoder
                self._report_progress(1.0, "Video processing complete!")
                return True
            except Exception as e:  # noqa: BLE001
                self._report_progress(1.0, f"Error: {str(e)}")
                print(f"Video processing error: {e}", file=sys.stderr)
                return False
        try:
            info = self.get_video_info(input_path)
            fps = info["fps"]
            self._report_progress(0.0, "Initializing video processing...")
            with tempfile.TemporaryDirectory() as tmp:
                tmp_dir = Path(tmp)
                self._report_progress(0.05, "Extracting frames...")
                pattern = str(tmp_dir / "frame_%05d.png")
                subprocess.run(["ffmpeg", "-i", input_path, "-qscale:v", "2", pattern], stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, check=True)
                frames = sorted(tmp_dir.glob("frame_*.png"))
                total = len(frames)
                if total == 0:
                    raise ValueError("No frames extracted from video")
                self._report_progress(0.1, f"Processing {total} frames...")
                method, max_size = (None, 64) if pixelize_func is None else pixelize_func
                failed, done = [], 0
                for lo in range(0, total, batch_size):
                    batch = frames[lo:lo + batch_size]
                    failed += self._process_batch(batch, ditherer, method, max_size, final_resize_multiplier)
                    done += len(batch)
                    self._report_progress(0.1 + 0.8 * (done / total), f"Processed {done}/{total} frames")
                if failed:
                    print(f"Fixing {len(failed)} failed frames by copying from nearest frames...", file=sys.stderr)
                    self._fix_failed_frames(failed, frames)
                self._report_progress(0.9, "Encoding final video...")
                subprocess.run(["ffmpeg", "-y", "-framerate", f"{fps:.5f}", "-i", pattern, "-i", input_path,
                                "-map", "0:v:0", "-map", "1:a?", "-map", "1:s?", "-c:v", "libx264", "-preset",
                                "medium", "-crf", "18", "-pix_fmt", "yuv420p", "-vframes", str(total), "-c:a", "copy",
                                "-c:s", "copy", output_path], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               check=True)
                self._report_progress(1.0, "Video processing complete!")
                return True
        except Exception as e:  # noqa: BLE001
            self._report_progress(1.0, f"Error: {str(e)}")
            print(f"Video processing error: {e}", file=sys.stderr)
            return False
