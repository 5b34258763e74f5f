"""A few Lloyd passes over the C4 image (33 M pixels, K = 32): the target of rocprofv3 passes."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import backend as be
g = torch.Generator(device='cuda'); g.manual_seed(99)
px = torch.randint(0, 256, (4320 * 7680, 3), dtype=torch.uint8, device='cuda', generator=g)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
c = torch.from_numpy(np.random.RandomState(1).rand(K, 3) * 255.0).cuda()
for _ in range(3):  # centres that look like k-means centres of this data
    s_, n_, _q = be.kmeans_step(px[::64].contiguous(), c)
    c = torch.where(n_[:, None] > 0, s_.double() / n_.clamp(min=1)[:, None].double(), c)
tot = torch.zeros(5 * K, dtype=torch.int64, device='cuda')
for _ in range(6): be.kmeans_step_into(px, c, tot, want_sq=False)
torch.cuda.synchronize(); print("done")
