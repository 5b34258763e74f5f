"""Error diffusion with palettes of 17..256 colours: the hierarchical <= 4-entry nearest table (EdTables::h4) against the 16^3 lists
(DP_ED_NO_H4=1), same process, experiments twin; one 4K frame, 24 frames, 256 frames; outputs compared.  usage: ed_h4_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["DITHER_PIE_EXPERIMENTS"] = "1"
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode

GLOBAL = not (len(sys.argv) > 1 and sys.argv[1] == "lds-only")   # sixteen-wave instances read the table from global memory
g = torch.Generator(device="cuda"); g.manual_seed(1)
frames = torch.randint(0, 256, (256, 2160, 3840, 3), dtype=torch.uint8, device="cuda", generator=g)
out = torch.empty_like(frames)


def t(d, n, reps=3):
    d.apply_dithering_frames(frames[:n], out=out[:n]); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); d.apply_dithering_frames(frames[:n], out=out[:n]); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for K, variant in ((256, "floyd_steinberg"), (64, "floyd_steinberg"), (32, "floyd_steinberg"), (256, "jjn"), (256, "atkinson")):
    pal = [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (K, 3))]
    d = ImageDitherer(K, DitherMode.ERROR_DIFFUSION, pal, False, {"variant": variant, "serpentine": "false"})
    res = {}
    for tag, env in (("lists", "1"), ("h4", None)):
        if env: os.environ["DP_ED_NO_H4"] = env
        else: os.environ.pop("DP_ED_NO_H4", None)
        if not GLOBAL: os.environ["DP_ED_H4_LDS_ONLY"] = "1"
        else: os.environ.pop("DP_ED_H4_LDS_ONLY", None)
        res[tag] = [t(d, n) for n in (1, 24, 256)]
        res[tag + "_hash"] = int(out[:24].to(torch.int64).sum().item())
    print(f"K={K:3d} {variant:16s} lists: 1 frame {res['lists'][0]:7.2f} ms  24: {res['lists'][1]:7.2f}  256: {res['lists'][2]:8.2f} | "
          f"h4: {res['h4'][0]:7.2f}  {res['h4'][1]:7.2f}  {res['h4'][2]:8.2f} | same bytes: {res['lists_hash'] == res['h4_hash']}", flush=True)
