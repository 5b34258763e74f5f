"""ColorReducer.reduce_colors on one 4K image of the bench content (smooth + grain, the 960x540 tile repeated 4x4), step by
step: PIL -> numpy, staging + H2D, distinct colours in first-occurrence order on the device (dp_distinct_first_u8), D2H, the
host's set-order replay + median cut (dp_median_cut_host).  Medians of 5."""
import sys; sys.path.insert(0, '.')
import time, ctypes as C
import numpy as np, torch
from PIL import Image
from dither_pie_amd import backend as be, _lib
from dither_pie_amd.dithering_lib import ColorReducer, _pinned_pair


def med(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2], r


rs = np.random.RandomState(3)
yy, xx = np.mgrid[0:540, 0:960]
img = np.clip(np.stack([80 + 60 * np.sin(xx / 300.0) + 40 * (yy / 540.0), 110 + 50 * np.cos(yy / 200.0) + 20 * np.sin(xx / 97.0),
                        160 + 70 * (yy / 540.0) + 10 * np.sin((xx + yy) / 50.0)], -1) + rs.normal(0, 3, (540, 960, 3)), 0, 255).astype(np.uint8)
pil = Image.fromarray(np.tile(img, (4, 4, 1)), "RGB")
ColorReducer.reduce_colors(pil, 256)
t_all, pal = med(lambda: ColorReducer.reduce_colors(pil, 256))
t_np, arr = med(lambda: np.array(pil.convert("RGB"), dtype=np.uint8).reshape(-1, 3))
n = len(arr)


def up():
    h_in, _ = _pinned_pair(3 * n)
    h_in[:3 * n].copy_(torch.from_numpy(arr).reshape(-1))
    t = h_in[:3 * n].cuda(non_blocking=True).view(n, 3)
    torch.cuda.synchronize()
    return t


t_up, t = med(up)
t_dev, d = med(lambda: be.distinct_first(t))
t_down, distinct = med(lambda: np.ascontiguousarray(d.cpu().numpy()))
L = _lib.load()
out = np.zeros((256, 3), np.int32); n_out = C.c_int(0)
t_cut, _ = med(lambda: L.dp_median_cut_host(distinct.ctypes.data, len(distinct), 8, out.ctypes.data, C.byref(n_out)))
order = np.empty(len(distinct), np.uint32); nd = C.c_int64(0)
t_set, _ = med(lambda: L.dp_pyset_order_host(distinct.ctypes.data, len(distinct), order.ctypes.data, C.byref(nd)))
print(f"reduce_colors(4K, 256): {t_all:.2f} ms   = PIL->numpy {t_np:.2f} + pinned staging and H2D {t_up:.2f} + distinct on the device {t_dev:.2f} "
      f"({len(distinct)} colours) + D2H {t_down:.2f} + host set order and cut {t_cut:.2f} (set order alone {t_set:.2f})")
