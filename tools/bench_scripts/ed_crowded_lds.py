import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else ".")
os.environ["DITHER_PIE_EXPERIMENTS"] = "1"
import numpy as np, torch
from PIL import Image
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
rs = np.random.RandomState(3)
yy, xx = np.mgrid[0:540, 0:960]
img = np.clip(np.stack([80 + 60 * np.sin(xx / 300.0) + 40 * (yy / 540.0), 110 + 50 * np.cos(yy / 200.0) + 20 * np.sin(xx / 97.0), 160 + 70 * (yy / 540.0) + 10 * np.sin((xx + yy) / 50.0)], -1) + rs.normal(0, 3, (540, 960, 3)), 0, 255).astype(np.uint8)
dark = np.clip(img.astype(np.int32) // 3 + rs.randint(0, 6, img.shape), 0, 255).astype(np.uint8)
def t(d, f, n, reps=3):
    o = torch.empty_like(f[:n])
    d.apply_dithering_frames(f[:n], out=o); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); d.apply_dithering_frames(f[:n], out=o); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
for nm, src in (("smooth", img), ("dark", dark)):
    f = torch.from_numpy(src).cuda().repeat(4, 4, 1).unsqueeze(0).repeat(24, 1, 1, 1).contiguous()
    for K in (32, 64, 128, 256):
        pal = ColorReducer.reduce_colors(Image.fromarray(src, "RGB"), K)
        d = ImageDitherer(K, DitherMode.ERROR_DIFFUSION, pal, False, {"variant": "floyd_steinberg", "serpentine": "false"})
        os.environ["DP_ED_NO_H4"] = "1"; a1, a24 = t(d, f, 1), t(d, f, 24)
        os.environ.pop("DP_ED_NO_H4"); b1, b24 = t(d, f, 1), t(d, f, 24)
        print(f"median cut {K:3d} of {nm:6s}: lists 1 frame {a1:7.2f} 24 frames {a24:7.2f} | table in LDS {b1:7.2f} {b24:7.2f}", flush=True)
