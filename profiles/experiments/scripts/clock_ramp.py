"""Duration of consecutive launches of the headline kernel (C2 batch) from a cold start: how long the clocks take to settle."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
from dither_pie_amd import backend as be
pal = [tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0, 256, (256, 3))]
f = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (24, 2160, 3840, 3), dtype=np.uint8)).cuda(); o = torch.empty_like(f)
d = ImageDitherer(256, DitherMode.BAYER, pal, False, {"size": "8x8"}).prepare()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
evs[0].record()
for i in range(n):
    d.apply_dithering_frames(f, out=o); evs[i + 1].record()
torch.cuda.synchronize()
ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
for a in range(0, n, 25):
    print(f"launches {a:4d}-{a+24:4d}: mean {sum(ts[a:a+25])/25:.4f} ms", flush=True)
