"""C4-shaped parity check: an 8K image dithered in 8 row bands with global coordinates equals the oracle on the whole image."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from oracle import oracle as orc
from dither_pie_amd import backend as be
orc.build()
arr = orc.rnd(4320, 7680, 99)
pal = orc.palr(32, 5)
P = be.Palette(*orc.prepare_palette(pal, False), accel=True)
ok = True
for mode, params in [("blue_noise", {"size": 64, "seed": 42}), ("bayer", {"size": "8x8"}), ("IGN", {"scale": 1.0, "seed": 3})]:
    ref = orc.apply_dithering(arr, pal, mode, params, False)
    outs = []
    for b in range(8):
        band = torch.from_numpy(arr[b * 540:(b + 1) * 540]).cuda()
        if mode == "blue_noise": o = be.ordered(band, P, be.MODE_MATRIX, thr=be.Thresholds.blue_noise(64, 42), y0=b * 540)
        elif mode == "bayer": o = be.ordered(band, P, be.MODE_MATRIX, thr=be.Thresholds.from_matrix(orc.bayer_matrix("8x8")), y0=b * 540)
        else: o = be.ordered(band, P, be.MODE_IGN, ign_scale=1.0, ign_seed=3, y0=b * 540)
        outs.append(o.cpu().numpy())
    same = np.array_equal(np.concatenate(outs), ref)
    print(mode, "8K in 8 bands == oracle:", same, flush=True)
    ok &= same
sys.exit(0 if ok else 1)
