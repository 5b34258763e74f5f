"""Would a zero-copy writer (vmsplice of the pinned output buffer into the encoder pipe: no kernel-side copy, no page allocation on
that side) speed up the READER of the overlapped pipe path, which is what bounds it?  Stand-in decoder -> Python reader, Python
writer -> stand-in encoder, together as two threads: write() against vmsplice().   python tools/bench_scripts/pipe_vmsplice.py"""
import ctypes, ctypes.util, fcntl, os, subprocess, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import pipe_standin as ps  # noqa: E402
N, H, W, BATCH = 600, 1080, 1920, 15
FB = H * W * 3
libc = ctypes.CDLL(ctypes.util.find_library("c"), use_errno=True)


class IoVec(ctypes.Structure):
    _fields_ = [("base", ctypes.c_void_p), ("len", ctypes.c_size_t)]


libc.vmsplice.argtypes = [ctypes.c_int, ctypes.POINTER(IoVec), ctypes.c_ulong, ctypes.c_uint]
libc.vmsplice.restype = ctypes.c_ssize_t


def widen(f):
    try:
        fcntl.fcntl(f.fileno(), 1031, 1 << 20)
    except Exception:  # noqa: BLE001
        pass


def reader(d, env, view, res):
    p = subprocess.Popen([os.path.join(d, "ffmpeg"), "-s", f"{W}x{H}", "pipe:1"], stdout=subprocess.PIPE, bufsize=0, env=env)
    widen(p.stdout)
    t = time.perf_counter(); total = 0
    while True:
        got = 0
        while got < len(view):
            n = p.stdout.readinto(view[got:])
            if not n: break
            got += n
        total += got
        if got < len(view): break
    res["rd"] = round(total / FB / (time.perf_counter() - t), 1); p.wait()


def writer(d, env, addr, nbytes, view, res, splice):
    p = subprocess.Popen([os.path.join(d, "ffmpeg"), "-s", f"{W}x{H}", "pipe:0", os.path.join(d, "o.bin")], stdin=subprocess.PIPE, bufsize=0, env=env)
    widen(p.stdin)
    fd = p.stdin.fileno()
    t = time.perf_counter(); left = N
    while left > 0:
        k = min(BATCH, left); todo = k * FB; off = 0
        if splice:
            while off < todo:
                iov = IoVec(addr + off, todo - off)
                n = libc.vmsplice(fd, ctypes.byref(iov), 1, 0)
                if n < 0:
                    res["err"] = os.strerror(ctypes.get_errno()); p.stdin.close(); p.wait(); return
                off += n
        else:
            mv = view[:todo]
            while len(mv):
                n = p.stdin.write(mv); mv = mv[n:]
        left -= k
    p.stdin.close(); p.wait()
    res["wr"] = round(N / (time.perf_counter() - t), 1)
    info, _ = ps.read_summary(os.path.join(d, "o.bin")); res["frames"] = info["frames"]


def main():
    import torch
    d = ps.build(tempfile.mkdtemp(prefix="dp_vs_"))
    env = ps.environment(d, N, H, W)
    a = torch.zeros(BATCH * FB, dtype=torch.uint8, pin_memory=True); b = torch.full((BATCH * FB,), 7, dtype=torch.uint8, pin_memory=True)
    va, vb = memoryview(a.numpy()), memoryview(b.numpy())
    for splice in (False, True, False, True):
        res = {}
        ta = threading.Thread(target=reader, args=(d, env, va, res)); tb = threading.Thread(target=writer, args=(d, env, b.data_ptr(), BATCH * FB, vb, res, splice))
        ta.start(); tb.start(); ta.join(); tb.join()
        print("vmsplice" if splice else "write   ", res, flush=True)
    res = {}; writer(d, env, b.data_ptr(), BATCH * FB, vb, res, True); print("vmsplice alone", res)
    res = {}; writer(d, env, b.data_ptr(), BATCH * FB, vb, res, False); print("write alone   ", res)


if __name__ == "__main__":
    main()
