"""Time of Palette(..., accel=True) for fresh 256-, 64- and 16-colour palettes (dp_palette_create + dp_palette_build_accel), and --
with a twin rebuilt with the stage timers of profiles/experiments/r05_priced_structures.md section 8 -- its stages on stderr."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import backend
from dither_pie_amd.dithering_lib import prepare_palette
for K, seed in ((256, 10), (256, 11), (256, 12), (64, 13), (16, 14), (1024, 15)):
    pal = [tuple(int(v) for v in c) for c in np.random.RandomState(seed).randint(0, 256, (K, 3))]
    torch.cuda.synchronize(); t = time.perf_counter()
    P = backend.Palette(*prepare_palette(pal, False), accel=True)
    torch.cuda.synchronize(); print("K=%4d seed %d: %.2f ms" % (K, seed, (time.perf_counter() - t) * 1e3), flush=True)
