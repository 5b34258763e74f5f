import time, numpy as np, torch
from PIL import Image
a = np.random.RandomState(0).randint(0, 256, (2160, 3840, 3), dtype=np.uint8)
im = Image.fromarray(a)
pin3 = torch.empty(2160*3840*3, dtype=torch.uint8).pin_memory(); host3 = pin3.numpy()
pin4 = torch.empty(2160*3840*4, dtype=torch.uint8).pin_memory(); host4 = pin4.numpy()
def enc_into(buf, raw):
    # ImageFile-style encoder writing straight into our buffer in chunks (no 25 MB bytes object)
    e = Image._getencoder(im.mode, "raw", raw); e.setimage(im.im, (0, 0) + im.size)
    pos = 0; bs = 1 << 22
    while True:
        l, s, d = e.encode(bs); buf[pos:pos+len(d)] = np.frombuffer(d, np.uint8); pos += len(d)
        if s: break
    return pos
V = {
 "copyto(pin3, frombuffer(tobytes()))": lambda: np.copyto(host3, np.frombuffer(im.tobytes(), dtype=np.uint8)),
 "tobytes() alone": lambda: im.tobytes(),
 "tobytes(raw RGBX) alone": lambda: im.tobytes("raw", "RGBX"),
 "copyto(pin4, frombuffer(tobytes(raw,RGBX)))": lambda: np.copyto(host4, np.frombuffer(im.tobytes("raw", "RGBX"), dtype=np.uint8)),
 "encoder chunks -> pin3": lambda: enc_into(host3, "RGB"),
 "encoder chunks -> pin4 (RGBX)": lambda: enc_into(host4, "RGBX"),
 "out: fromarray(host3)": lambda: Image.fromarray(host3.reshape(2160, 3840, 3), "RGB"),
 "out: frombuffer(RGBX view of pin4).copy()": lambda: Image.frombuffer("RGB", (3840, 2160), host4, "raw", "RGBX", 0, 1).copy(),
 "out: frombytes(RGB, raw RGBX, host4)": lambda: Image.frombytes("RGB", (3840, 2160), host4.tobytes(), "raw", "RGBX"),
}
ts = {k: [] for k in V}
for rep in range(10):
    for k, fn in V.items():
        t = time.perf_counter(); fn(); ts[k].append((time.perf_counter() - t) * 1e3)
for k, v in ts.items():
    v.sort(); print(f"{k:48s} median {v[len(v)//2]:6.2f} ms  min {v[0]:6.2f}")
assert enc_into(host3, "RGB") == 2160*3840*3 and np.array_equal(host3.reshape(2160,3840,3), a)
