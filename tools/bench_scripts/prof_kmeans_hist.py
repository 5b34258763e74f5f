"""The k-means fit over the colour histogram on the C4 image (33 M pixels, K = 32): histogram build + a few passes, the
target of rocprofv3 passes (kernel trace / PMC).  usage: prof_kmeans_hist.py [K] [noise|smooth]"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import backend as be
g = torch.Generator(device='cuda'); g.manual_seed(99)
kind = sys.argv[2] if len(sys.argv) > 2 else "noise"
N = 4320 * 7680
if kind == "noise":
    px = torch.randint(0, 256, (N, 3), dtype=torch.uint8, device='cuda', generator=g)
else:
    yy, xx = torch.meshgrid(torch.arange(4320, device='cuda'), torch.arange(7680, device='cuda'), indexing='ij')
    px = torch.stack([(xx * 255 // 7679), (yy * 255 // 4319), ((xx + yy) * 255 // (7679 + 4319))], -1).to(torch.int16).reshape(-1, 3)
    px = (px + torch.randint(-6, 7, px.shape, device='cuda', generator=g).to(torch.int16)).clamp(0, 255).to(torch.uint8).contiguous()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hist = be.ColourHistogram(px)
c = torch.from_numpy(np.random.RandomState(1).rand(K, 3) * 255.0).cuda()
for _ in range(3):  # centres that look like k-means centres of this data
    s_, n_, _q = hist.step(c)
    c = torch.where(n_[:, None] > 0, s_.double() / n_.clamp(min=1)[:, None].double(), c).contiguous()
tot = torch.zeros(5 * K, dtype=torch.int64, device='cuda')
for _ in range(2): hist.add(px, accumulate=False)
for _ in range(6): hist.step_into(c, tot, False)
torch.cuda.synchronize(); print("done")
