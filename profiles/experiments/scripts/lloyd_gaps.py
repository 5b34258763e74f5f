"""One Lloyd fit over the 8K image (K = 32): the target of a rocprofv3 --kernel-trace run whose timestamps show how much of an
iteration is kernels and how much is gaps between them."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import kmeans
g = torch.Generator(device='cuda'); g.manual_seed(99)
px = torch.randint(0, 256, (4320 * 7680, 3), dtype=torch.uint8, device='cuda', generator=g)
sample = kmeans.seed_sample(px, px.shape[0], 0, 42)
init = kmeans.kmeans_plusplus(sample, 32, np.random.RandomState(42))
for _ in range(2):
    res = kmeans.lloyd(px, init)
torch.cuda.synchronize(); print("iterations", res[2])
