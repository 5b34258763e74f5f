#!/usr/bin/env python3
"""bench.py -- dither+quantize throughput of the MI355X backend (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Without a launcher (no WORLD_SIZE in the environment) and N > 1, this process starts the N ranks itself through
torch.distributed.run BEFORE it touches the GPU, relays rank 0's JSON line and exits with the launcher's code.

Workload (config.workload = "C2"): Bayer 8x8 + 256-colour nearest-palette on 3840x2160 RGB frames
(BASELINE.json configs[1]); one step = one pass of ImageDitherer.apply_dithering_frames over a batch of
24 distinct synthetic 4K frames (597 MB in, 597 MB out: larger than the 256 MiB Infinity Cache) that are
already resident in HBM.  With N > 1 every rank runs the same batch on its own GPU (frames shard with no
collective; weak scaling) and `value` is the whole-job pixel rate = N * pixels / max-over-ranks time.

One JSON line is printed by rank 0.  `roofline` is for the dominant kernel (ordered pass 1), timed with
HIP events on its own stream inside the library (dp_profile_*); `cpu_baseline` is the CPU oracle
(oracle/, the checker -- never the product path) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H4K, W4K = 2160, 3840
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy rate
BYTES_PER_PX = 6       # algorithmic: 3 B read + 3 B written per pixel


def palr(K, seed=7):
    return [tuple(int(v) for v in c) for c in np.random.RandomState(seed).randint(0, 256, (K, 3))]


def make_frames(torch, n, h, w, dev, first_seed=None, numpy_frames=0, gen_seed=1234):
    """n synthetic uint8 frames in HBM: i.i.d. uniform bytes.  The first `numpy_frames` of them are SURVEY 8(d)'s
    rnd(h, w, first_seed + i) (numpy legacy RNG, generated on the host); the others come from torch's device generator
    (same distribution, no host time)."""
    g = torch.Generator(device=dev)
    g.manual_seed(gen_seed)
    frames = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device=dev, generator=g)
    if first_seed is not None:
        for i in range(min(n, max(1, numpy_frames))):
            kat = np.random.RandomState(first_seed + i).randint(0, 256, (h, w, 3), dtype=np.uint8)
            frames[i].copy_(torch.from_numpy(kat))
    return frames


def kernel_sources_sha16():
    """Identity of the kernels a PMC file was taken with: sha256 over the ordered-path sources."""
    import hashlib
    hsh = hashlib.sha256()
    for name in ("ordered.hip", "accel.hip", "dp_internal.h", "tree_query.hip.h"):
        path = os.path.join(ROOT, "dither_pie_amd", "csrc", name)
        if os.path.exists(path):
            with open(path, "rb") as f:
                hsh.update(f.read())
    return hsh.hexdigest()[:16]


def sources_sha16(names):
    import hashlib
    hsh = hashlib.sha256()
    for name in names:
        path = os.path.join(ROOT, "dither_pie_amd", "csrc", name)
        if os.path.exists(path):
            with open(path, "rb") as f:
                hsh.update(f.read())
    return hsh.hexdigest()[:16]


_LEG_SOURCES = {"c2_crowded": ("ordered.hip", "accel.hip", "dp_internal.h", "tree_query.hip.h"),
                "c2_use_gamma": ("ordered.hip", "accel.hip", "dp_internal.h", "tree_query.hip.h"),
                "c4_kmeans_pass": ("kmeans_hist.hip", "kmeans_label.hip.h", "wave_util.hip.h"),
                "c4_kmeans_histogram": ("kmeans_hist.hip", "kmeans_label.hip.h", "wave_util.hip.h"),
                "c3": ("ediff.hip", "ed_nearest.hip.h", "dp_internal.h"),
                "c5_video": ("ordered.hip", "accel.hip", "dp_internal.h", "tree_query.hip.h")}


PMC_LEGS = "r05_pmc_legs.json"


def leg_traffic(result, name, pmc_leg=None):
    """HBM traffic of a leg's dominant kernel, ESTIMATED from the committed PMC passes of this round (profiles/r05_pmc_legs.json:
    FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes of the tools/bench_scripts/*_prof.py workload named there -- for c3 and the
    histogram build that workload is the leg's own launch shape): the measured ratio to that workload's algorithmic bytes applied to
    this leg's.  Nothing is counted during this run (PMC counters cannot be read from inside the process), hence the key's name;
    reported only while the kernel sources are the ones the passes were taken with."""
    leg = result.get(name)
    path = os.path.join(ROOT, "profiles", PMC_LEGS)
    if not leg or not os.path.exists(path):
        return
    try:
        with open(path) as f:
            rec = json.load(f)["legs"].get(pmc_leg or name)
        if not rec:
            return
        if rec.get("kernel_sources_sha16") != sources_sha16(_LEG_SOURCES[name]):
            leg["traffic_estimated_from_profile"] = None
            leg["traffic_source"] = f"profiles/{PMC_LEGS} was taken with other kernel sources: not reported"
            return
        ratio = rec["derived"].get("traffic_over_algorithmic")
        if ratio:
            leg["traffic_estimated_from_profile"] = int(ratio * leg["algorithmic_bytes"])
            leg["traffic_over_algorithmic"] = round(ratio, 3)
            leg["traffic_source"] = (f"profiles/{PMC_LEGS} [" + (pmc_leg or name) + "]: FETCH_SIZE x2 + WRITE_SIZE per launch of "
                                     + rec["workload"] + ", as a ratio to that workload's algorithmic bytes (a committed profile of "
                                     "the same kernel sources, not a counter read in this run)")
    except Exception:  # noqa: BLE001
        return


def rocprof_average_ms(kernel_substring, stats_name="r05_kernel_stats.csv"):
    """AverageNs of a kernel in the committed rocprofv3 --kernel-trace --stats summary of `bench.py --no-extra --no-cpu-baseline`
    (profiles/r05_kernel_stats.csv: every launch of the run, the slow first ones under the profiler included) -> ms, or None."""
    import csv
    path = os.path.join(ROOT, "profiles", stats_name)
    if not os.path.exists(path):
        return None
    try:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if kernel_substring in row.get("Name", ""):
                    return float(row["AverageNs"]) / 1e6
    except Exception:  # noqa: BLE001
        return None
    return None


def cpu_baseline(n_frames=16):
    """The CPU oracle (C restatement, OpenMP over rows) on a bounded sample of the same workload."""
    from oracle import oracle as orc
    orc.build()
    pal = orc.palr(256)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = orc.set_threads(min(avail, 16))  # the CPU share of one GPU on the bench box is 16 cores
    orc.apply_dithering(orc.rnd(16, W4K, 1), pal, "bayer", {"size": "8x8"})  # warm-up (OpenMP pool, page-in)
    px, dt = 0, 0.0
    for i in range(n_frames):
        arr = orc.rnd(H4K, W4K, 1234 + i)
        t0 = time.perf_counter()
        orc.apply_dithering(arr, pal, "bayer", {"size": "8x8"})
        dt += time.perf_counter() - t0
        px += arr.shape[0] * arr.shape[1]
    res = {"value": round(px / dt / 1e6, 3), "unit": "Mpixel/s", "cores": threads, "kind": "port",
           "os_cpu_count": os.cpu_count(), "cpus_available": avail,
           "sample": f"{n_frames} full 3840x2160 frames rnd(2160,3840,1234+i), Bayer 8x8, 256 colours; C oracle "
                     f"(scipy-order KD-tree query per pixel), OpenMP over rows on {threads} threads, {dt:.2f} s"}
    # the same port on ONE thread (SURVEY 8d leg a), two frames
    orc.set_threads(1)
    t0 = time.perf_counter()
    for i in range(2):
        orc.apply_dithering(orc.rnd(H4K, W4K, 1234 + i), pal, "bayer", {"size": "8x8"})
    dt1 = time.perf_counter() - t0
    orc.set_threads(threads)
    res["single_thread"] = {"value": round(2 * H4K * W4K / dt1 / 1e6, 3), "unit": "Mpixel/s", "cores": 1, "kind": "port",
                            "sample": f"2 full 4K frames, the same C oracle on one thread, {dt1:.2f} s"}
    # the reference's video parallelism (video_processor.py:43-45, 321-322): multiprocessing.Pool(min(4, cores-1)) over
    # frames, here 1080p Bayer 4x4 / 16 uniform colours (C5), each worker running the single-threaded port on its frames
    try:
        import multiprocessing as mp
        workers = min(4, max(1, (os.cpu_count() or 2) - 1))
        nfr = 4 * workers
        with mp.get_context("fork").Pool(processes=workers) as pool:
            pool.map(_c5_frame_worker, range(workers))  # warm-up: library load in every worker
            t0 = time.perf_counter()
            pool.map(_c5_frame_worker, range(nfr))
            dtp = time.perf_counter() - t0
        res["video_pool"] = {"value": round(nfr / dtp, 2), "unit": "1080p frames/s", "cores": workers, "kind": "port",
                             "sample": f"{nfr} frames rnd(1080,1920,i), Bayer 4x4, 16 uniform colours, Pool({workers}) of "
                                       f"single-threaded C-oracle workers (no PNG/ffmpeg I/O), {dtp:.2f} s"}
    except Exception as e:  # noqa: BLE001
        res["video_pool"] = {"error": str(e)}
    try:  # the same math stated with the reference's own third-party calls (scipy KDTree.query(k=2) + numpy)
        arr = orc.rnd(H4K, W4K, 1234)
        t0 = time.perf_counter()
        out = orc.ordered_scipy(arr, pal, orc.bayer_matrix("8x8"), False, workers=threads)
        dt2 = time.perf_counter() - t0
        import hashlib
        res["alt"] = {"value": round(H4K * W4K / dt2 / 1e6, 3), "unit": "Mpixel/s", "cores": threads, "kind": "port",
                      "sample": f"1 frame, scipy.spatial.KDTree.query(k=2, workers={threads}) + single-threaded numpy "
                                f"post-processing as in dithering_lib.py:355-378, {dt2:.2f} s",
                      "kat_ok": hashlib.sha256(out.tobytes()).hexdigest()[:16] == "7041bd52fdea90b5"}
    except Exception as e:  # noqa: BLE001  (scipy missing on the box)
        res["alt"] = {"error": str(e)}
    return res


def _c5_frame_worker(i):
    from oracle import oracle as orc
    orc.set_threads(1)
    out = orc.apply_dithering(orc.rnd(1080, 1920, i), orc.generate_uniform_palette(16), "bayer", {"size": "4x4"})
    return int(out[0, 0, 0])


def pipes_leg(torch, dev, ditherer, n_frames=600, batch=15, h=1080, w=1920):
    """c5_pipes: what the unchanged CLI / GUI waits on for a video -- VideoProcessor.process_video_streaming (the drop-in of
    video_processor.py:172-390) end to end: decoder pipe -> pinned slots -> HBM -> kernels -> pinned slots -> encoder pipe, three
    stages overlapped on rotating slots.  There is no ffmpeg in the image: the decoder and encoder are tools/pipe_standin.c
    (compiled here with gcc; they stream from / to memory, no codec), so the number is the plumbing's, not ffmpeg's.  Beside it:
    the bare pipes (stand-in -> /dev/null, stand-in -> a Python reader that discards, a Python writer -> stand-in), the serial
    loop of rounds 2-4 on the same stream, and a check of every output byte through the encoder's checksum."""
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pipe_standin as ps
    from dither_pie_amd import video_processor as vproc
    if shutil.which("gcc") is None:
        return {"error": "no gcc on the box: the stand-in decoder / encoder could not be built"}
    tmp = tempfile.mkdtemp(prefix="dp_pipes_")
    saved = {k: os.environ.get(k) for k in ("PATH", "DP_STANDIN_W", "DP_STANDIN_H", "DP_STANDIN_FRAMES", "DP_STANDIN_DISTINCT", "DP_STANDIN_KEEP")}
    try:
        d = ps.build(os.path.join(tmp, "bin"))
        distinct, keep = 8, 2
        env = ps.environment(d, n_frames, h, w, distinct=distinct, keep=keep)
        os.environ.update({k: env[k] for k in saved})
        fb = h * w * 3
        # ---- the bare pipes -------------------------------------------------------------------------------------------
        t0 = time.perf_counter()
        with open(os.devnull, "wb") as nul:
            subprocess.check_call([os.path.join(d, "ffmpeg"), "-s", f"{w}x{h}", "pipe:1"], stdout=nul, env=env)
        devnull_fps = n_frames / (time.perf_counter() - t0)
        stage = torch.empty(batch * fb, dtype=torch.uint8, pin_memory=True)
        view = memoryview(stage.numpy())

        def read_ceiling(view, res=None):
            p = subprocess.Popen([os.path.join(d, "ffmpeg"), "-s", f"{w}x{h}", "pipe:1"], stdout=subprocess.PIPE, bufsize=0, env=env)
            vproc.VideoProcessor._widen_pipe(p.stdout)
            t = time.perf_counter()
            total = 0
            while True:
                got = 0
                while got < len(view):
                    n = p.stdout.readinto(view[got:])
                    if not n:
                        break
                    got += n
                total += got
                if got < len(view):
                    break
            dt = time.perf_counter() - t
            p.stdout.close()
            p.wait()
            if res is not None:
                res["read"] = total / fb / dt
            return total / fb / dt

        def write_ceiling(view, res=None, name="ceiling.bin"):
            p = subprocess.Popen([os.path.join(d, "ffmpeg"), "-s", f"{w}x{h}", "pipe:0", os.path.join(tmp, name)],
                                 stdin=subprocess.PIPE, bufsize=0, env=env)
            vproc.VideoProcessor._widen_pipe(p.stdin)
            t = time.perf_counter()
            left = n_frames
            while left > 0:
                k = min(batch, left)
                vproc.VideoProcessor._write_all(p.stdin, view[:k * fb])
                left -= k
            p.stdin.close()
            p.wait()
            if res is not None:
                res["write"] = n_frames / (time.perf_counter() - t)
            return n_frames / (time.perf_counter() - t)

        read_fps = max(read_ceiling(view) for _ in range(2))
        write_fps = max(write_ceiling(view) for _ in range(2))
        # both pipes at once (two bare threads, no queues, no GPU): what the machine gives the two directions TOGETHER -- each
        # runs slower than alone (page allocation and memory traffic of two kernel-side copies), and that is the pipeline's ceiling
        import threading
        stage2 = torch.empty(batch * fb, dtype=torch.uint8, pin_memory=True)
        view2 = memoryview(stage2.numpy())
        both = []
        for _ in range(3):
            got2 = {}
            ta = threading.Thread(target=read_ceiling, args=(view2, got2))
            tb = threading.Thread(target=write_ceiling, args=(view, got2, "ceiling2.bin"))
            ta.start()
            tb.start()
            ta.join()
            tb.join()
            both.append(got2)
        both_best = max(both, key=lambda g: min(g["read"], g["write"]))
        # ---- the pipeline ---------------------------------------------------------------------------------------------
        out_path = os.path.join(tmp, "out.bin")
        vp = vproc.VideoProcessor(devices=[dev.index])

        def one(overlap):
            info = vp.get_video_info("standin.mp4")
            t = time.perf_counter()
            done = vp._stream_through_pipes("standin.mp4", out_path, ditherer, None, 64, batch, None, info, overlap=overlap)
            dt = time.perf_counter() - t
            return done / dt, dict(vp.last_pipe_stats)

        one(True)  # warm-up: pinned slots faulted in, the stream's workspace allocated
        runs = [one(True) for _ in range(3)]
        fps, stats = max(runs, key=lambda r: r[0])
        summary, kept = ps.read_summary(out_path)
        serial_fps, serial_stats = one(False)
        # the public entry point (probe + milestones included), once
        t = time.perf_counter()
        ok_public = vp.process_video_streaming("standin.mp4", out_path, ditherer, None, batch_size=batch)
        public_fps = n_frames / (time.perf_counter() - t)
        # ---- every output byte: the encoder's order-sensitive checksum against the kernels on the same frames in HBM -----
        base = torch.from_numpy(ps.frames(distinct, h, w, distinct=distinct)).to(dev)
        idx = torch.arange(n_frames, device=dev)
        expect, kept_ok = 0, True
        for lo in range(0, n_frames, 100):
            sel = idx[lo:lo + 100]
            fr = base[sel % distinct].clone()
            tags = sel.to(torch.int64)
            for b in range(4):
                fr.view(len(sel), -1)[:, b] = ((tags >> (8 * b)) & 255).to(torch.uint8)
            o = ditherer.apply_dithering_frames(fr)
            sums = o.view(len(sel), -1).sum(dim=1, dtype=torch.int64).tolist()
            for j, sv in enumerate(sums):
                expect = (expect + (lo + j + 1) * sv) & 0xffffffffffffffff
            if lo == 0:
                kept_ok = bool((o[:keep].cpu().numpy() == kept).all())
        busy = {k: round(stats[k], 4) for k in ("read_s", "gpu_submit_s", "gpu_wait_s", "write_s", "wall_s")}
        slower = min(read_fps, write_fps)
        return {"metric": "1080p frames/s end to end through rawvideo pipes (decode pipe -> GPU -> encode pipe), Bayer 4x4 + 16 uniform colours",
                "frames": n_frames, "batch_frames": batch, "slots": stats.get("slots"), "frames_per_s": round(fps, 1),
                "frames_per_s_runs": [round(r[0], 1) for r in runs],
                "stage_busy_s": busy,
                "stage_busy_note": "per run of frames_per_s: read_s = reader thread inside read() on the decoder pipe; gpu_submit_s = this thread "
                                   "queueing H2D + kernels + D2H (no device wait); gpu_wait_s = writer thread waiting for a batch's event; "
                                   "write_s = writer thread inside write() on the encoder pipe; the stages overlap, wall_s is the call",
                "pipe_ceiling": {"decoder_to_devnull_fps": round(devnull_fps, 1), "decoder_pipe_to_pinned_buffer_fps": round(read_fps, 1),
                                 "pinned_buffer_to_encoder_pipe_fps": round(write_fps, 1),
                                 "both_pipes_at_once_fps": {k: round(v, 1) for k, v in both_best.items()},
                                 "note": "stand-in decoder -> /dev/null; stand-in decoder -> a Python loop that reads into one pinned buffer and "
                                         "discards; a Python loop writing one pinned buffer -> stand-in encoder; and the last two running "
                                         "TOGETHER as two bare threads (no queues, no GPU) -- the two kernel-side copies slow each other down.  The "
                                         "pipeline itself hands its output slots to the encoder pipe by reference (vmsplice: no copy on "
                                         "that side), so it can pass the both-at-once pair and approach the decoder pipe alone"},
                "frac_of_slower_pipe": round(fps / slower, 3),
                "frac_of_slower_pipe_with_both_running": round(fps / min(both_best.values()), 3),
                "serial_loop_frames_per_s": round(serial_fps, 1),
                "public_call_frames_per_s": round(public_fps, 1), "public_call_ok": bool(ok_public),
                "bytes_ok": bool(summary["frames"] == n_frames and summary["bytes"] == n_frames * fb and summary["wsum"] == expect and kept_ok),
                "cores": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(),
                "note": "NO real ffmpeg was available: decoder and encoder are compiled stand-ins (tools/pipe_standin.c) that stream "
                        f"{distinct} distinct frames round-robin from memory / read and checksum to memory -- the plumbing's rate, an upper "
                        "bound for a video whose codec keeps up; a libx264 encode of 1080p is far slower than any number here"}
    except Exception as e:  # noqa: BLE001 - a host-path leg must not take the headline down
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        shutil.rmtree(tmp, ignore_errors=True)


def spawn_ranks(n, argv):
    """Parent of a self-launched multi-GPU run: never touches the GPU; starts n ranks through torch.distributed.run
    (127.0.0.1 rendezvous on a free port), passes everything through and relays the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=24, help="4K frames per batch (per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary 1080p / error-diffusion lines")
    ap.add_argument("--spawn", action="store_true", help="start the ranks through torch.distributed.run even for --gpus 1")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="(testing the N > 1 code path on a one-GPU box) every rank uses cuda:0 and the collectives run over "
                         "gloo; the numbers of such a run mean nothing")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        raise SystemExit(spawn_ranks(args.gpus, [a for a in sys.argv[1:] if a != "--spawn"]))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match the launcher's WORLD_SIZE={os.environ['WORLD_SIZE']}")

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ  # under torchrun even a single rank uses RCCL
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner to stdout when the first communicator comes up: keep stdout for the one
        # JSON line by pointing fd 1 at stderr until that has happened
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.rehearse_on_one_gpu:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from dither_pie_amd import backend
    from dither_pie_amd.dithering_lib import DitherMode, ImageDitherer

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    step_ms = []  # per-step durations of the last timed() call with per_step=True (events on the launch stream)

    def timed(fn, steps, warmup, per_step=False):
        for _ in range(warmup):
            fn()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if per_step else None
        barrier()
        t0 = time.perf_counter()
        if per_step:
            evs[0].record()
        for i in range(steps):
            fn()
            if per_step:
                evs[i + 1].record()
        barrier()
        dt = time.perf_counter() - t0
        if per_step:
            step_ms[:] = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
        if distributed:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def kernel_ms(fn, reps=3):
        """mean duration of the main kernel of one call of fn (HIP events recorded by the library on the launch stream:
        dp_profile_*; the fix-up pass / repair launch separately) -> (main ms, second-pass ms) per call"""
        fn()
        torch.cuda.synchronize()
        backend.profile_enable(True)
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        m, f2, n = backend.profile_read()
        backend.profile_enable(False)
        return m / reps, f2 / reps

    def leg(kernel, ms, alg_bytes, **more):
        """one structured bench leg: the kernel that dominates it, its duration per call and its algorithmic bytes against the
        HBM roofline (6 B per dithered pixel, 3 B per pixel of a Lloyd pass)"""
        gbs = alg_bytes / (ms * 1e-3) / 1e9
        return {"kernel": kernel, "kernel_ms": round(ms, 4), "algorithmic_bytes": int(alg_bytes), "achieved_gbs": round(gbs, 1),
                "frac": round(gbs / HBM_PEAK_GBS, 4), **more}

    # ---------------- headline: C2 --------------------------------------------------------------
    pal256 = palr(256)
    # (a video keeps its palette for every frame: the search accelerator is prepared up front, outside the timed region;
    # what it costs is reported as extra.accel_build_ms)
    dith = ImageDitherer(256, DitherMode.BAYER, pal256, False, {"size": "8x8"}).prepare()
    frames = make_frames(torch, args.frames, H4K, W4K, dev, first_seed=1234, numpy_frames=args.frames)
    out = torch.empty_like(frames)
    px_per_step = args.frames * H4K * W4K

    def step():
        dith.apply_dithering_frames(frames, out=out)

    # one-off per palette, outside the timed region (a video reuses it for every frame): KD-tree + cell table + tie codes
    torch.cuda.synchronize()
    t_acc = time.perf_counter()
    step()
    torch.cuda.synchronize()
    first_call_ms = (time.perf_counter() - t_acc) * 1e3
    import hashlib
    kat_ok = hashlib.sha256(out[0].cpu().numpy().tobytes()).hexdigest()[:16] == "7041bd52fdea90b5"

    # untimed: bring the GPU clocks up and fault in every buffer before the contract's W warm-up steps, so that a
    # short run (few steps) measures the same thing as a long one
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    backend.profile_enable(True)
    dt = timed(step, args.steps, args.warmup, per_step=True)
    main_ms, fix_ms, launches = backend.profile_read()
    backend.profile_enable(False)
    per_step = sorted(step_ms)
    # warm-up launches are in the record too: average per launch is what matters
    k_ms = main_ms / max(launches, 1)
    value = world * px_per_step * args.steps / dt / 1e6
    achieved = BYTES_PER_PX * px_per_step / (k_ms * 1e-3) / 1e9
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the figure
    # comes from the committed rocprofv3 --pmc pass of this same command (profiles/pmc_pass.sh)
    # (profiles/pmc_pass.sh); the file names the kernel sources it was taken with and is ignored when they have changed
    traffic, traffic_src, valu_per_launch = None, None, None
    pmc_name = "r05_pmc_ordered.json"
    pmc_file = os.path.join(ROOT, "profiles", pmc_name)
    if args.frames == 24 and os.path.exists(pmc_file):
        try:
            with open(pmc_file) as f:
                pmc = json.load(f)
            if pmc.get("kernel_sources_sha16") == kernel_sources_sha16():
                traffic = int(pmc["derived"]["hbm_traffic_bytes_per_launch"])
                traffic_src = f"profiles/{pmc_name} (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes, same kernel sources)"
                valu_per_launch = pmc["derived"].get("valu_wave_instructions_per_launch")
            else:
                traffic_src = f"profiles/{pmc_name} was taken with other kernel sources: not reported"
        except Exception:  # noqa: BLE001
            traffic = None
    result = {
        "metric": "Mpixel/s dither+quantize @4K 256-color (Bayer 8x8 + nearest-palette, uint8 RGB in/out)",
        "value": round(value, 1), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "ms_per_step_min": round(per_step[0], 4), "ms_per_step_median": round(per_step[len(per_step) // 2], 4),
        "rccl_world_size": dist.get_world_size() if distributed else 1,
        **({"rehearsal": "all ranks on cuda:0, gloo collectives: code-path check only, the numbers mean nothing"}
           if args.rehearse_on_one_gpu else {}),
        "config": {"workload": "C2: Bayer 8x8 + 256-colour nearest palette palr(256,7), 3840x2160 RGB, "
                               f"{args.frames} distinct frames rnd(2160,3840,1234+i) per GPU resident in HBM, frames sharded per rank",
                   "frames_per_gpu": args.frames, "h": H4K, "w": W4K, "colors": 256, "matrix": "8x8",
                   "parallelism": f"frames x{world}"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "ordered_lean_kernel<1,8> (main pass of dp_ordered_u8)", "kernel_ms": round(k_ms, 4),
                     "fixup_ms": round(fix_ms / max(launches, 1), 4), "launches": launches,
                     "algorithmic_bytes_per_launch": BYTES_PER_PX * px_per_step},
        "parity_kat_4k": bool(kat_ok),
        "first_call_ms": round(first_call_ms, 2),
    }
    # the same fraction from the committed rocprofv3 summary of this command (its average over ALL launches of the run, the slow
    # first ones included): the least favourable of the three ways to time the kernel, reported next to `frac`
    rp_ms = rocprof_average_ms("ordered_lean_kernel<1, 8, false, false, false>") if args.frames == 24 else None
    if rp_ms:
        try:   # the same run launch by launch (profiles/r05_kernel_trace_headline.csv): its median does not move with one preempted launch
            import csv
            with open(os.path.join(ROOT, "profiles", "r05_kernel_trace_headline.csv"), newline="") as f:
                durs = sorted(float(r["duration_us"]) for r in csv.DictReader(f))
            med_ms = durs[len(durs) // 2] / 1e3
            result["roofline"]["rocprof_median_kernel_ms"] = round(med_ms, 4)
            result["roofline"]["frac_rocprof_median"] = round(BYTES_PER_PX * px_per_step / (med_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        except Exception:  # noqa: BLE001
            pass
        result["roofline"]["rocprof_avg_kernel_ms"] = round(rp_ms, 4)
        result["roofline"]["frac_rocprof_avg"] = round(BYTES_PER_PX * px_per_step / (rp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        result["roofline"]["frac_rocprof_avg_source"] = "profiles/r05_kernel_stats.csv (AverageNs of the kernel over every launch of `bench.py --no-extra --no-cpu-baseline` under rocprofv3 --kernel-trace --stats)"
    if valu_per_launch:
        # what actually bounds the kernel: wave64 VALU instructions of one launch (same PMC file) against
        # the issue rate of 1024 SIMDs, one instruction per 4 cycles, at the 2.4 GHz peak clock
        # profiles/microbench/valu_rate_results.txt: these integer operations issue once per ~4.3 cycles per SIMD
        result["roofline"]["valu_issue_frac"] = round(valu_per_launch * 4.3 / (1024 * 2.4e9 * k_ms * 1e-3), 4)
        result["roofline"]["valu_wave_instructions_per_launch"] = int(valu_per_launch)

    # ---------------- C5 (the metric's second half): 1000 x 1080p frames, Bayer 4x4, 16 uniform colours -------------
    # STRONG scaling: the 1000 frames are split into contiguous blocks over the ranks, frames resident in
    # HBM, compute only; frames/s = 1000 / max-over-ranks time of one pass.
    from dither_pie_amd.dithering_lib import ColorReducer
    total = 1000
    lo, hi = rank * total // world, (rank + 1) * total // world
    d5 = ImageDitherer(16, DitherMode.BAYER, ColorReducer.generate_uniform_palette(16), False, {"size": "4x4"}).prepare()
    # ALL of this rank's frames are resident in HBM and distinct ("1000 frames pre-generated per GPU, contiguous blocks
    # of 1000/G frames per GPU"; 6.2 GB in + 6.2 GB out at one rank): the first `n5np` of the block are rnd(1080, 1920, seed) with
    # seeds lo .. lo + n5np - 1 (numpy legacy RNG on the host, ~3 s per 100), the rest come from the device generator exactly as
    # make_frames does for C2 (same distribution, no host time).  One pass = one walk over all of them, in equal launches of at
    # most 100 frames (1000 -> 10 x 100; 125 at eight ranks -> 63 + 62).
    n5r = hi - lo
    n5np = min(100, n5r)
    f5 = make_frames(torch, max(1, n5r), 1080, 1920, dev, first_seed=lo, numpy_frames=n5np, gen_seed=1234 + lo)
    o5 = torch.empty_like(f5)
    k5 = max(1, -(-n5r // 100))
    cuts5 = [n5r * i // k5 for i in range(k5 + 1)]

    def video_pass():
        for a5, b5 in zip(cuts5, cuts5[1:]):
            if b5 > a5:
                d5.apply_dithering_frames(f5[a5:b5], out=o5[a5:b5])

    n5 = 40  # passes between the two barriers: at 8 ranks a pass is ~0.6 ms, an RCCL barrier ~0.1 ms (2.5 % at 10 passes, 0.6 % at 40)
    t5 = timed(video_pass, n5, 2)
    n5f = max(1, cuts5[1] - cuts5[0])
    k5_ms, _ = kernel_ms(lambda: d5.apply_dithering_frames(f5[:n5f], out=o5[:n5f]))
    result["c5_video"] = {"metric": "1080p frames/s, Bayer 4x4 + 16 uniform colours, 1000 frames", "scaling": "strong",
                          "frames_total": total, "n_gpus": world, "frames_per_s": round(total * n5 / t5, 1), "passes": n5,
                          "frames_this_rank": n5r, "launches_per_pass": k5, "frames_per_launch": [b5 - a5 for a5, b5 in zip(cuts5, cuts5[1:])],
                          **leg(f"ordered_lean_kernel<1,4,HALF> (one launch per block of {n5f} frames)", k5_ms, BYTES_PER_PX * n5f * 1080 * 1920),
                          "note": f"contiguous blocks of 1000/N frames per rank, no collective; all {n5r} frames of this rank's block distinct and "
                                  f"resident in HBM ({2 * n5r * 1080 * 1920 * 3 / 1e9:.1f} GB in + out), one pass walks every one of them once: "
                                  f"frames {lo}..{lo + n5np - 1} are rnd(1080,1920,seed=frame number) of SURVEY 8(d) (numpy legacy RNG), the other "
                                  f"{n5r - n5np} come from torch's device generator (same distribution), as make_frames does for C2"}
    leg_traffic(result, "c5_video")
    del f5, o5
    if world == 1 and not args.no_extra:
        result["c5_pipes"] = pipes_leg(torch, dev, d5)

    # ---------------- secondary lines (same process, after the headline) -------------------------
    if not args.no_extra:
        extra = {}
        try:   # (a secondary leg that fails must not take the headline line down with it)
            # what the first call on a new palette pays once: KD-tree + 16^3 cell table + tie codes of all 2^24 colours
            from dither_pie_amd.dithering_lib import prepare_palette
            torch.cuda.synchronize()
            t_b = time.perf_counter()
            pal_obj = backend.Palette(*prepare_palette(palr(256, 11), False), accel=True)
            torch.cuda.synchronize()
            extra["accel_build_ms"] = round((time.perf_counter() - t_b) * 1e3, 2)
            extra["accel_build_note"] = ("dp_palette_create + dp_palette_build_accel for a fresh 256-colour palette; outside the timed "
                                         "region (a video keeps its palette), but it is what ONE 4K image pays on top of the kernel")
            del pal_obj
            # on-box streaming copy of the same 597 MB batch (3 B read + 3 B written per pixel, like the kernel)
            tc = timed(lambda: out.copy_(frames), 10, 2) / 10
            copy_gbs = BYTES_PER_PX * px_per_step / tc / 1e9
            result["roofline"]["measured_copy_gbs"] = round(copy_gbs, 1)
            result["roofline"]["frac_of_measured_copy"] = round(achieved / copy_gbs, 4)
            # C2 on the structured, tie-rich frame of SURVEY 8(d): every frame of the batch = grad(2160, 3840)
            yy, xx = torch.meshgrid(torch.arange(H4K, device=dev), torch.arange(W4K, device=dev), indexing="ij")
            gradf = torch.stack([xx % 256, yy % 256, ((xx + yy) // 2) % 256], -1).to(torch.uint8)
            fg = gradf.unsqueeze(0).expand(args.frames, -1, -1, -1).contiguous()
            del yy, xx, gradf
            tg = timed(lambda: dith.apply_dithering_frames(fg, out=out), 5, 1) / 5
            extra["c2_grad_frames_mpixel_per_s"] = round(world * px_per_step / tg / 1e6, 1)
            extra["c2_grad_note"] = "same kernel on 24 copies of the structured frame grad(2160,3840) (0.36 % tied pixels)"
            del fg
            # C2-shaped, but content and palette as a user has them: smooth image-like frames (gradients + grain) and the
            # 256-colour median-cut palette of that very content -- the palette crowds into the cells where the pixels are
            from PIL import Image
            from dither_pie_amd.dithering_lib import ColorReducer as _CR
            rs = np.random.RandomState(3)
            yy, xx = np.mgrid[0:540, 0:960]
            img = np.clip(np.stack([80 + 60 * np.sin(xx / 300.0) + 40 * (yy / 540.0), 110 + 50 * np.cos(yy / 200.0) + 20 * np.sin(xx / 97.0),
                                    160 + 70 * (yy / 540.0) + 10 * np.sin((xx + yy) / 50.0)], -1) + rs.normal(0, 3, (540, 960, 3)), 0, 255).astype(np.uint8)
            pal_mc = _CR.reduce_colors(Image.fromarray(img, "RGB"), 256)
            fi = torch.from_numpy(img).to(dev).repeat(4, 4, 1).unsqueeze(0).repeat(args.frames, 1, 1, 1).contiguous()
            dmc = ImageDitherer(256, DitherMode.BAYER, pal_mc, False, {"size": "8x8"}).prepare()
            dmc.apply_dithering_frames(fi, out=out)
            ti = timed(lambda: dmc.apply_dithering_frames(fi, out=out), 3, 1) / 3
            extra["c2_image_like_median_cut256_mpixel_per_s"] = round(world * px_per_step / ti / 1e6, 1)
            extra["c2_image_like_note"] = ("smooth frames with grain + the 256-colour median-cut palette of that content: the palette "
                                           "crowds a few cells of the colour cube (one-byte-per-entry table over warped cells, the whole "
                                           "octree in LDS, every pixel resolved in place: ordered_compact_kernel)")
            kmc_ms, kmc_fix = kernel_ms(lambda: dmc.apply_dithering_frames(fi, out=out))
            result["c2_crowded"] = leg("ordered_compact_kernel<1,WARP,HALF>", kmc_ms, BYTES_PER_PX * px_per_step, fixup_ms=round(kmc_fix, 4),
                                       workload="C2-shaped: image-like frames + their own median-cut 256 palette, Bayer 8x8")
            # what the palette itself costs (the reference's default palette source, once per image / per video): median cut of a
            # 4K image of that content, distinct colours found on the GPU, CPython's set order replayed natively
            img4k = Image.fromarray(fi[0].cpu().numpy(), "RGB")
            _CR.reduce_colors(img4k, 256)
            mc_ts = []
            for _ in range(3):
                t_m = time.perf_counter()
                _CR.reduce_colors(img4k, 256)
                mc_ts.append((time.perf_counter() - t_m) * 1e3)
            extra["median_cut256_4k_ms"] = round(sorted(mc_ts)[1], 1)
            extra["median_cut_note"] = ("ColorReducer.reduce_colors(4K image, 256), median of 3: host side (dp_median_cut_host) after a GPU pass "
                                        "for the distinct colours; the reference needs seconds for this step")
            del img4k
            # the same content with the reference's default palette size: 16 colours by median cut
            pal_mc16 = _CR.reduce_colors(Image.fromarray(img, "RGB"), 16)
            dmc16 = ImageDitherer(16, DitherMode.BAYER, pal_mc16, False, {"size": "8x8"}).prepare()
            dmc16.apply_dithering_frames(fi, out=out)
            ti16 = timed(lambda: dmc16.apply_dithering_frames(fi, out=out), 3, 1) / 3
            extra["c2_image_like_median_cut16_mpixel_per_s"] = round(world * px_per_step / ti16 / 1e6, 1)
            del fi, yy, xx
            # C2 with use_gamma=True (float32 palette coordinates, pixels through lut_in): the float cell table
            dgam = ImageDitherer(256, DitherMode.BAYER, pal256, True, {"size": "8x8"}).prepare()
            dgam.apply_dithering_frames(frames, out=out)
            tgm = timed(lambda: dgam.apply_dithering_frames(frames, out=out), 3, 1) / 3
            extra["c2_use_gamma_mpixel_per_s"] = round(world * px_per_step / tgm / 1e6, 1)
            kg_ms, kg_fix = kernel_ms(lambda: dgam.apply_dithering_frames(frames, out=out))
            result["c2_use_gamma"] = leg("ordered_compact_float_kernel<2>", kg_ms, BYTES_PER_PX * px_per_step, fixup_ms=round(kg_fix, 4),
                                         workload="C2 with use_gamma=True (float32 palette coordinates, pixels through lut_in)")
            # what the unchanged CLI / GUI sees per image: apply_dithering(PIL) on one 4K image, host -> device -> host included
            pil_img = Image.fromarray(frames[0].cpu().numpy(), "RGB")
            dith.apply_dithering(pil_img)
            pil_ts = []
            for _ in range(10):
                t_p = time.perf_counter()
                dith.apply_dithering(pil_img)
                pil_ts.append((time.perf_counter() - t_p) * 1e3)
            extra["pil_4k_ms"] = round(sorted(pil_ts)[len(pil_ts) // 2], 2)
            extra["pil_4k_note"] = ("ImageDitherer.apply_dithering(PIL image) on one 3840x2160 image, median of 10: PIL -> pinned host "
                                    "buffer -> HBM -> kernel -> pinned host buffer -> PIL (PCIe both ways; the kernel share is ~0.02 ms)")
            del out
            from dither_pie_amd.dithering_lib import ColorReducer
            # C3: Floyd-Steinberg, 16 colours, 4K, a batch of frames (one wave per frame)
            nf3 = 8 if args.rehearse_on_one_gpu else 256   # (a rehearsal puts every rank's buffers on ONE GPU)
            d3 = ImageDitherer(16, DitherMode.ERROR_DIFFUSION, ColorReducer.generate_uniform_palette(16), False,
                               {"variant": "floyd_steinberg", "serpentine": "false"})
            f3 = frames[:min(nf3, args.frames)].repeat((nf3 + args.frames - 1) // args.frames, 1, 1, 1)[:nf3]
            o3 = torch.empty_like(f3)
            t3 = min(timed(lambda: d3.apply_dithering_frames(f3, out=o3), 2, 1 - i) for i in range(2)) / 2   # (the better of two pairs)
            extra["c3_fs_k16_4k_mpixel_per_s"] = round(world * nf3 * H4K * W4K / t3 / 1e6, 2)
            k3_ms, k3_rep = kernel_ms(lambda: d3.apply_dithering_frames(f3, out=o3), 2)
            result["c3"] = leg("ed_wavefront_kernel (one workgroup of 16 waves per frame)", k3_ms, BYTES_PER_PX * nf3 * H4K * W4K,
                               frames_in_flight=nf3, bound="dependency chain of the raster scan (W + skew*H steps per frame), not HBM",
                               workload="C3: Floyd-Steinberg, 16 uniform colours, 3840x2160")
            # the C3 batch as SURVEY 8(d) words it (the 24 frames of C2), and one frame: few frames in flight, each frame's
            # bands spread over several workgroups
            o24 = o3[:args.frames]
            d3.apply_dithering_frames(frames, out=o24)
            t24 = timed(lambda: d3.apply_dithering_frames(frames, out=o24), 2, 1) / 2
            extra["c3_fs_k16_4k_24_frames_ms"] = round(t24 * 1e3, 2)
            t1f = timed(lambda: d3.apply_dithering_frames(frames[:1], out=o24[:1]), 2, 1) / 2
            extra["c3_fs_k16_4k_one_frame_ms"] = round(t1f * 1e3, 2)
            extra["c3_note"] = (f"{nf3} frames in flight per GPU (one workgroup of 16 waves per frame), "
                                "bit-exact float32 error accumulation")
            # a batch larger than the device (what a video is): one persistent workgroup per CU whose waves run on into the next
            # frame's bands, so that the tail of a frame (34 bands over 16 waves: the third round has two waves busy) is not idle
            nf3l = 3 * nf3
            f3l = f3.repeat(3, 1, 1, 1)
            o3l = torch.empty_like(f3l)
            t3l = timed(lambda: d3.apply_dithering_frames(f3l, out=o3l), 2, 1) / 2
            extra["c3_fs_k16_4k_long_batch_mpixel_per_s"] = round(world * nf3l * H4K * W4K / t3l / 1e6, 2)
            extra["c3_long_batch_note"] = (f"{nf3l} frames per GPU in one call: persistent workgroups, three frames each")
            del f3l, o3l
            # the same frame with 256 (random) colours: candidate lists instead of a 16-entry table
            pal3b = [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (256, 3))]
            d3b = ImageDitherer(256, DitherMode.ERROR_DIFFUSION, pal3b, False, {"variant": "floyd_steinberg", "serpentine": "false"})
            d3b.apply_dithering_frames(frames[:1], out=o24[:1])
            t1b = timed(lambda: d3b.apply_dithering_frames(frames[:1], out=o24[:1]), 2, 1) / 2
            extra["c3_fs_k256_4k_one_frame_ms"] = round(t1b * 1e3, 2)
            if not args.rehearse_on_one_gpu:   # the same palette, 256 frames in flight (the hierarchical nearest table read from L2)
                f3b = frames.repeat((256 + args.frames - 1) // args.frames, 1, 1, 1)[:256]
                o3b = torch.empty_like(f3b)
                d3b.apply_dithering_frames(f3b, out=o3b)
                t3b = min(timed(lambda: d3b.apply_dithering_frames(f3b, out=o3b), 2, 0) for _ in range(2)) / 2   # (the better of two pairs: one preempted 22 ms launch once read as 65 Gpx/s)
                extra["c3_fs_k256_4k_mpixel_per_s"] = round(world * 256 * H4K * W4K / t3b / 1e6, 2)
                del f3b, o3b
            # C4: k-means 32-colour palette from a 7680x4320 image + blue-noise dither, the image split into row
            # bands over the ranks; the per-iteration exchange is one RCCL all-reduce of [32,5] int64
            from dither_pie_amd import kmeans, sharding
            del f3, o3
            lo4, hi4 = sharding.shard_range(4320, rank, world)
            # SURVEY 8(d)'s C4 image rnd(4320, 7680, 99) (numpy legacy RNG on the host, ~1 s), this rank's row band of it
            band = torch.from_numpy(np.random.RandomState(99).randint(0, 256, (4320, 7680, 3), dtype=np.uint8)[lo4:hi4]).to(dev).contiguous()

            def c4(src=None):
                src = band if src is None else src
                pal, _, _, iters = kmeans.fit_palette(src.reshape(-1, 3), 32, 42, n_total=4320 * 7680, offset=lo4 * 7680)
                d4 = ImageDitherer(32, DitherMode.BLUE_NOISE, pal, False, {"size": 64, "seed": 42})
                sharding.dither_band(d4, src, lo4)
                return iters

            iters4 = c4()
            t4 = min(timed(c4, 1, 0) for _ in range(3))   # (one fit + dither per measurement; the best of three)
            extra["c4_8k_kmeans32_plus_blue_noise_seconds"] = round(t4, 4)
            # the same on image-like content (smooth gradients + grain): what a photograph looks like to the fit
            yy4, xx4 = torch.meshgrid(torch.arange(lo4, hi4, device=dev), torch.arange(7680, device=dev), indexing="ij")
            g4 = torch.Generator(device=dev)
            g4.manual_seed(99)
            smooth = torch.stack([(xx4 * 255 // 7679), (yy4 * 255 // 4319), ((xx4 + yy4) * 255 // (7679 + 4319))], -1).to(torch.int16)
            smooth = (smooth + torch.randint(-6, 7, smooth.shape, device=dev, generator=g4).to(torch.int16)).clamp(0, 255).to(torch.uint8).contiguous()
            del yy4, xx4
            iters4s = c4(smooth)
            t4s = min(timed(lambda: c4(smooth), 1, 0) for _ in range(3))
            extra["c4_8k_image_like_kmeans32_plus_blue_noise_seconds"] = round(t4s, 4)
            # the fit's two kernels, each against its own bytes: the histogram build reads the pixels once (3 B/px); a Lloyd
            # iteration reads 16 KB per occupied 16^3 cell of the colour cube (64 MB when all 4096 are occupied), not the pixels
            from dither_pie_amd import backend as _be
            c4c = torch.from_numpy(np.random.RandomState(1).rand(32, 3) * 255.0).to(dev)
            tot4 = torch.zeros(160, dtype=torch.int64, device=dev)
            bpx = band.reshape(-1, 3)
            n4 = bpx.numel() // 3
            hist4 = _be.ColourHistogram(bpx)
            kb_ms, _ = kernel_ms(lambda: hist4.add(bpx, accumulate=False), 3)
            info4 = hist4.buf[1 << 26:].view(torch.int32)
            occ4 = int(info4[4096].item())
            result["c4_kmeans_histogram"] = leg("hist_count / plan / scatter / parts kernels (pixels -> count[colour] over 2^24 colours by "
                                                "partition, once per fit; kernel_ms = the four together)", kb_ms, 3 * n4,
                                                bound="LDS atomics and scattered 2-byte stores of the partition (pixels read twice, 2 B/px of "
                                                      "buckets written and read, 64 MB of table): not HBM",
                                                workload=f"C4: the {n4} pixels of this rank's band of rnd(4320,7680,99) ({world} band(s))")
            hist4.step_into(c4c, tot4, False)
            kp_ms, _ = kernel_ms(lambda: hist4.step_into(c4c, tot4, False), 5)
            tp4 = timed(lambda: hist4.step_into(c4c, tot4, False), 10, 2) / 10
            result["c4_kmeans_pass"] = leg("hist_pass_kernel<false,false> (one Lloyd pass over the colour histogram, candidate lists built in-kernel)",
                                           kp_ms, occ4 * 16384, occupied_cells=occ4, pass_ms_with_memset=round(tp4 * 1e3, 4),
                                           iterations_of_the_fit=int(iters4),
                                           workload=f"C4: one Lloyd pass, K=32, over the histogram of this rank's band of the 7680x4320 image ({world} band(s)); "
                                                    "algorithmic bytes = 16 KB per occupied cell")
            # the pass over the PIXELS (what fits with more than 256 clusters, or fewer than 2^19 pixels, still run): 3 B/px
            _be.kmeans_step_into(bpx, c4c, tot4, want_sq=False)
            kx_ms, _ = kernel_ms(lambda: _be.kmeans_step_into(bpx, c4c, tot4, want_sq=False), 5)
            result["c4_kmeans_pixel_pass"] = leg("kmeans_cells_kernel<false,2> (its list build launch not included)", kx_ms, 3 * n4,
                                                 workload="the same pass over the pixels themselves (round 3's Lloyd pass)")
            extra["c4_kmeans_pass_ms"] = round(tp4 * 1e3, 4)
            extra["c4_note"] = (f"rnd(4320,7680,99) in {world} row band(s); per rank: pixels -> colour histogram once, then Lloyd over the "
                                f"histogram ({iters4} iterations on noise, {iters4s} on the image-like content; one launch per iteration on one "
                                "rank, pass / int64 all-reduce / update when sharded), blue-noise(64,42) dither of the band with global coordinates")
            del hist4, smooth
            for _leg in ("c2_crowded", "c2_use_gamma", "c3", "c4_kmeans_pass", "c4_kmeans_histogram"):
                leg_traffic(result, _leg)
        except Exception as e:  # noqa: BLE001
            import traceback
            extra["error"] = f"{type(e).__name__}: {e}"
            print(f"bench.py: a secondary leg failed on rank {rank}:\n{traceback.format_exc()}", file=sys.stderr)
        result["extra"] = extra

    if rank == 0 and not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
