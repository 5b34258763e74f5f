"""When should the sixteen-wave diffusion instances read the hierarchical nearest table from L2 (and when are the 8^3 lists better)?
Batched Floyd-Steinberg (256 4K frames) both ways for random / uniform / median-cut / k-means-like palettes on noise and on image-like
frames, next to the table's size and its mean depth at the palette's own colours (DP_ED_H4_REPORT).  usage: ed_h4_global_policy.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["DITHER_PIE_EXPERIMENTS"] = "1"; os.environ["DP_ED_H4_REPORT"] = "1"
import numpy as np, torch
from PIL import Image
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
rs = np.random.RandomState(3)
yy, xx = np.mgrid[0:540, 0:960]
img = np.clip(np.stack([80 + 60 * np.sin(xx / 300.0) + 40 * (yy / 540.0), 110 + 50 * np.cos(yy / 200.0) + 20 * np.sin(xx / 97.0), 160 + 70 * (yy / 540.0) + 10 * np.sin((xx + yy) / 50.0)], -1) + rs.normal(0, 3, (540, 960, 3)), 0, 255).astype(np.uint8)
dark = np.clip(img.astype(np.int32) // 3 + rs.randint(0, 6, img.shape), 0, 255).astype(np.uint8)
g = torch.Generator(device="cuda"); g.manual_seed(1)
content = {"noise": torch.randint(0, 256, (64, 2160, 3840, 3), dtype=torch.uint8, device="cuda", generator=g),
           "smooth": torch.from_numpy(img).cuda().repeat(4, 4, 1).unsqueeze(0).repeat(64, 1, 1, 1).contiguous(),
           "dark": torch.from_numpy(dark).cuda().repeat(4, 4, 1).unsqueeze(0).repeat(64, 1, 1, 1).contiguous()}
out = torch.empty((256, 2160, 3840, 3), dtype=torch.uint8, device="cuda")
def t(d, f, reps=2):
    f4 = f.repeat(4, 1, 1, 1)
    d.apply_dithering_frames(f4, out=out); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); d.apply_dithering_frames(f4, out=out); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
cases = []
for K in (32, 64, 128, 256):
    cases.append((f"random {K}", [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (K, 3))], "noise"))
    cases.append((f"median cut {K} of smooth", ColorReducer.reduce_colors(Image.fromarray(img, "RGB"), K), "smooth"))
    cases.append((f"median cut {K} of dark", ColorReducer.reduce_colors(Image.fromarray(dark, "RGB"), K), "dark"))
for name, pal, what in cases:
    d = ImageDitherer(len(pal), DitherMode.ERROR_DIFFUSION, pal, False, {"variant": "floyd_steinberg", "serpentine": "false"})
    os.environ.pop("DP_ED_H4_LDS_ONLY", None)
    os.environ["DP_ED_H4_GLOBAL"] = "1"   # (force it: the library itself reads the table from L2 only where it is shallow)
    a = t(d, content[what])
    os.environ.pop("DP_ED_H4_GLOBAL", None)
    os.environ["DP_ED_H4_LDS_ONLY"] = "1"
    b = t(d, content[what])
    print(f"{name:28s} on {what:6s}: table from L2 {a:8.2f} ms   lists {b:8.2f} ms   L2/lists {a / b:.3f}", flush=True)
