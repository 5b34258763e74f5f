import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else ".")
os.environ["DITHER_PIE_EXPERIMENTS"] = "1"
import numpy as np, torch
from PIL import Image
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
rs = np.random.RandomState(3)
yy, xx = np.mgrid[0:540, 0:960]
img = np.clip(np.stack([80 + 60 * np.sin(xx / 300.0) + 40 * (yy / 540.0), 110 + 50 * np.cos(yy / 200.0) + 20 * np.sin(xx / 97.0), 160 + 70 * (yy / 540.0) + 10 * np.sin((xx + yy) / 50.0)], -1) + rs.normal(0, 3, (540, 960, 3)), 0, 255).astype(np.uint8)
frames = torch.from_numpy(img).cuda().repeat(4, 4, 1).unsqueeze(0).repeat(256, 1, 1, 1).contiguous()
out = torch.empty_like(frames)
def t(d, n, reps=3):
    d.apply_dithering_frames(frames[:n], out=out[:n]); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); d.apply_dithering_frames(frames[:n], out=out[:n]); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
pal = ColorReducer.reduce_colors(Image.fromarray(img, "RGB"), 256)
d = ImageDitherer(256, DitherMode.ERROR_DIFFUSION, pal, False, {"variant": "floyd_steinberg", "serpentine": "false"})
ref = None
for label, env in (("lists only", {"DP_ED_NO_H4": "1"}), ("h4 in LDS / global batched", {}), ("h4 too big for LDS -> lists (1, 24), global batched", {"DP_ED_H4_LDS_WORDS": "20000"}),
                   ("h4 LDS, batched lists", {"DP_ED_H4_LDS_ONLY": "1"})):   # (a fifth variant, the few-frames instances reading a table larger than LDS from L2, lost and cost the LDS path its ds_read addressing: removed from the kernel)
    for k in ("DP_ED_NO_H4", "DP_ED_H4_LDS_WORDS", "DP_ED_H4_LDS_ONLY"): os.environ.pop(k, None)
    os.environ.update(env)
    r = [t(d, n) for n in (1, 24, 256)]
    h = int(out[:24].to(torch.int64).sum().item())
    ref = h if ref is None else ref
    print(f"{label:55s} 1 frame {r[0]:7.2f}  24: {r[1]:7.2f}  256: {r[2]:8.2f}  same {h == ref}", flush=True)
