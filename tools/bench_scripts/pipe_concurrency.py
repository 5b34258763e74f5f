"""Why do the reader and the writer of the overlapped pipe path each run at half their stand-alone rate when they run together?
Stand-in decoder -> Python reader and Python writer -> stand-in encoder: alone, together as two threads of one process, together
as two processes; pinned (hipHostMalloc) and pageable buffers.  python tools/bench_scripts/pipe_concurrency.py"""
import multiprocessing as mp
import os
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pipe_standin as ps  # noqa: E402

N, H, W, BATCH = 600, 1080, 1920, 15
FB = H * W * 3


def widen(f):
    import fcntl
    try:
        fcntl.fcntl(f.fileno(), 1031, 1 << 20)
    except Exception:  # noqa: BLE001
        pass


def reader(d, env, view, res, key="rd"):
    p = subprocess.Popen([os.path.join(d, "ffmpeg"), "-s", f"{W}x{H}", "pipe:1"], stdout=subprocess.PIPE, bufsize=0, env=env)
    widen(p.stdout)
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
    res[key] = round(total / FB / (time.perf_counter() - t), 1)
    p.wait()


def writer(d, env, view, res, key="wr"):
    p = subprocess.Popen([os.path.join(d, "ffmpeg"), "-s", f"{W}x{H}", "pipe:0", os.path.join(d, f"o_{key}.bin")], stdin=subprocess.PIPE,
                         bufsize=0, env=env)
    widen(p.stdin)
    t = time.perf_counter()
    left = N
    while left > 0:
        k = min(BATCH, left)
        mv = view[:k * FB]
        while len(mv):
            n = p.stdin.write(mv)
            mv = mv[n:]
        left -= k
    p.stdin.close()
    p.wait()
    res[key] = round(N / (time.perf_counter() - t), 1)


def proc_entry(which, d, env, q):
    import numpy as np
    buf = np.zeros(BATCH * FB, np.uint8)
    res = {}
    (reader if which == "rd" else writer)(d, env, memoryview(buf), res, which)
    q.put(res)


def main():
    import numpy as np
    import torch
    d = ps.build(tempfile.mkdtemp(prefix="dp_pc_"))
    env = ps.environment(d, N, H, W)
    print("cpus", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?")
    for kind in ("pageable", "pinned"):
        if kind == "pinned" and not torch.cuda.is_available():
            continue
        mk = (lambda: torch.zeros(BATCH * FB, dtype=torch.uint8, pin_memory=True).numpy()) if kind == "pinned" else (lambda: np.zeros(BATCH * FB, np.uint8))
        a, b = memoryview(mk()), memoryview(mk())
        for rep in range(2):
            res = {}
            reader(d, env, a, res)
            writer(d, env, b, res)
            alone = dict(res)
            res = {}
            ta = threading.Thread(target=reader, args=(d, env, a, res))
            tb = threading.Thread(target=writer, args=(d, env, b, res))
            ta.start(); tb.start(); ta.join(); tb.join()
            print(kind, "alone", alone, "two threads", res, flush=True)
    q = mp.get_context("spawn").Queue()
    ps_ = [mp.get_context("spawn").Process(target=proc_entry, args=(w, d, env, q)) for w in ("rd", "wr")]
    for p in ps_:
        p.start()
    out = {}
    for _ in ps_:
        out.update(q.get())
    for p in ps_:
        p.join()
    print("two processes (pageable)", out)
    # two readers in two threads (is it the GIL, or the direction?)
    a, b = memoryview(np.zeros(BATCH * FB, np.uint8)), memoryview(np.zeros(BATCH * FB, np.uint8))
    res = {}
    ta = threading.Thread(target=reader, args=(d, env, a, res, "rd1"))
    tb = threading.Thread(target=reader, args=(d, env, b, res, "rd2"))
    ta.start(); tb.start(); ta.join(); tb.join()
    print("two readers, two threads", res)


if __name__ == "__main__":
    main()
