"""Host-side conversions of one 3840x2160 RGB image, interleaved medians (GPU-box host): PIL -> pinned staging buffer, staging buffer -> PIL.
usage: pil_probe.py"""
import time, numpy as np, torch
from PIL import Image
import PIL
a = np.random.RandomState(0).randint(0, 256, (2160, 3840, 3), dtype=np.uint8)
im = Image.fromarray(a)
pin = torch.empty(2160 * 3840 * 3, dtype=torch.uint8).pin_memory() if torch.cuda.is_available() else torch.empty(2160 * 3840 * 3, dtype=torch.uint8)
host = pin.numpy()
variants_in = {
    "copyto(pin, frombuffer(tobytes()))": lambda: np.copyto(host, np.frombuffer(im.tobytes(), dtype=np.uint8)),
    "copyto(pin, asarray(im).reshape(-1))": lambda: np.copyto(host, np.asarray(im).reshape(-1)),
    "tobytes() alone": lambda: im.tobytes(),
    "asarray(im) alone": lambda: np.asarray(im),
}
variants_out = {
    "frombuffer(RGB).copy()": lambda: Image.frombuffer("RGB", (3840, 2160), host, "raw", "RGB", 0, 1).copy(),
    "frombuffer(RGB) (lazy view)": lambda: Image.frombuffer("RGB", (3840, 2160), host, "raw", "RGB", 0, 1),
    "fromarray(host.reshape)": lambda: Image.fromarray(host.reshape(2160, 3840, 3), "RGB"),
    "frombytes(RGB, bytes(host))": lambda: Image.frombytes("RGB", (3840, 2160), host.tobytes()),
}
for group in (variants_in, variants_out):
    ts = {k: [] for k in group}
    for rep in range(12):
        for k, fn in group.items():
            t = time.perf_counter(); fn(); ts[k].append((time.perf_counter() - t) * 1e3)
    for k, v in ts.items():
        v.sort(); print(f"{k:40s} median {v[len(v) // 2]:6.2f} ms   min {v[0]:6.2f}", flush=True)
print("Pillow", PIL.__version__, "numpy", np.__version__)
