"""Where the overlapped pipe path loses time against the bare concurrent pipes: the same stream with an identity batch function
(no GPU), with the real kernels, at several batch sizes and slot counts.  python tools/bench_scripts/pipe_stages.py"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pipe_standin as ps  # noqa: E402

N, H, W = 600, 1080, 1920


def main():
    import torch
    from dither_pie_amd import video_processor as v
    from dither_pie_amd.dithering_lib import ColorReducer, DitherMode, ImageDitherer
    d = ps.build(tempfile.mkdtemp(prefix="dp_pst_"))
    os.environ.update({k: val for k, val in ps.environment(d, N, H, W).items() if k == "PATH" or k.startswith("DP_STANDIN_")})
    d5 = ImageDitherer(16, DitherMode.BAYER, ColorReducer.generate_uniform_palette(16), False, {"size": "4x4"}).prepare()
    out = os.path.join(d, "out.bin")

    def one(label, batch, slots, run=None, overlap=True):
        vp = v.VideoProcessor(devices=[0])
        vp.PIPE_SLOTS = slots
        info = vp.get_video_info("x.mp4")
        best = None
        for _ in range(3):
            t = time.perf_counter()
            vp._stream_through_pipes("x.mp4", out, d5, None, 64, batch, None, info, run=run, overlap=overlap)
            dt = time.perf_counter() - t
            st = vp.last_pipe_stats
            if best is None or dt < best[0]:
                best = (dt, dict(st))
        dt, st = best
        print(f"{label:34s} batch {batch:3d} slots {slots}: {N / dt:7.1f} fps  read {st['read_s']:.3f} submit {st['gpu_submit_s']:.3f} "
              f"wait {st['gpu_wait_s']:.3f} write {st['write_s']:.3f} wall {st['wall_s']:.3f}", flush=True)

    ident = lambda x: x  # noqa: E731
    one("identity (no GPU), overlapped", 15, 3, ident)
    one("identity (no GPU), serial", 15, 3, ident, overlap=False)
    one("kernels, overlapped", 15, 3)
    one("kernels, serial", 15, 3, overlap=False)
    for batch in (2, 4, 8, 30, 60):
        one("kernels, overlapped", batch, 3)
    for slots in (2, 4, 6):
        one("kernels, overlapped", 15, slots)
    one("identity (no GPU), overlapped", 4, 4, ident)


if __name__ == "__main__":
    main()
