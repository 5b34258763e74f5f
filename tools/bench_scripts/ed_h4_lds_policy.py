"""The few-frames error-diffusion instances: hierarchical table in LDS (kept for EVERY palette: DP_ED_H4_ALWAYS=1) against the
16^3 / 8^3 lists (DP_ED_NO_H4=1), per palette with the table's depth at the palette's colours (DP_ED_H4_REPORT=1, stderr) --
the measurement behind the limit in build_ed_cells (ediff.hip).  Re-taken after the table walk got its ds_read addressing back."""
import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else ".")
os.environ["DITHER_PIE_EXPERIMENTS"] = "1"
os.environ["DP_ED_H4_ALWAYS"] = "1"
os.environ["DP_ED_H4_REPORT"] = "1"
import numpy as np, torch
from PIL import Image
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
rs = np.random.RandomState(3)
yy, xx = np.mgrid[0:540, 0:960]
img = np.clip(np.stack([80 + 60 * np.sin(xx / 300.0) + 40 * (yy / 540.0), 110 + 50 * np.cos(yy / 200.0) + 20 * np.sin(xx / 97.0), 160 + 70 * (yy / 540.0) + 10 * np.sin((xx + yy) / 50.0)], -1) + rs.normal(0, 3, (540, 960, 3)), 0, 255).astype(np.uint8)
dark = np.clip(img.astype(np.int32) // 3 + rs.randint(0, 6, img.shape), 0, 255).astype(np.uint8)
mid = np.clip(img.astype(np.int32) // 2 + 40 + rs.randint(0, 4, img.shape), 0, 255).astype(np.uint8)
def t(d, f, n, reps=3):
    o = torch.empty_like(f[:n])
    d.apply_dithering_frames(f[:n], out=o); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); d.apply_dithering_frames(f[:n], out=o); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best, int(o.to(torch.int64).sum().item())
def row(label, pal, f):
    K = len(pal)
    sys.stderr.write(f"## {label}\n"); sys.stderr.flush()
    d = ImageDitherer(K, DitherMode.ERROR_DIFFUSION, pal, False, {"variant": "floyd_steinberg", "serpentine": "false"})
    os.environ["DP_ED_NO_H4"] = "1"; (a1, h1), (a24, _) = t(d, f, 1), t(d, f, 24)
    os.environ.pop("DP_ED_NO_H4"); (b1, h2), (b24, _) = t(d, f, 1), t(d, f, 24)
    print(f"{label:28s}: lists 1 frame {a1:7.2f} 24 frames {a24:7.2f} | table in LDS {b1:7.2f} {b24:7.2f} | {100 * (b1 / a1 - 1):+6.1f} % {100 * (b24 / a24 - 1):+6.1f} %  same {h1 == h2}", flush=True)
for nm, src in (("smooth", img), ("mid", mid), ("dark", dark)):
    f = torch.from_numpy(src).cuda().repeat(4, 4, 1).unsqueeze(0).repeat(24, 1, 1, 1).contiguous()
    for K in (32, 64, 128, 256):
        row(f"median cut {K:3d} of {nm}", ColorReducer.reduce_colors(Image.fromarray(src, "RGB"), K), f)
f = torch.from_numpy(img).cuda().repeat(4, 4, 1).unsqueeze(0).repeat(24, 1, 1, 1).contiguous()
for K in (24, 32, 64, 128, 256):
    row(f"random {K:3d} (seed 7)", [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (K, 3))], f)
