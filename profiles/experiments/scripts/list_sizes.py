"""How long are the exact candidate lists T(x)-unions per cell for the headline palette at cell sizes 16, 8 and 4?
(design probe: would 4 candidates per 8^3 cell do?)"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pal = torch.tensor(np.random.RandomState(7).randint(0, 256, (K, 3)), dtype=torch.int32, device='cuda')
pp = (pal * pal).sum(1)
for cs in (16, 8, 4):
    n = 256 // cs
    sizes = torch.zeros(n * n * n, dtype=torch.int32, device='cuda')
    ax = torch.arange(256, device='cuda', dtype=torch.int32)
    # process one r-slab of cells at a time
    for rc in range(n):
        r = ax[rc * cs:(rc + 1) * cs]
        x = torch.stack(torch.meshgrid(r, ax, ax, indexing='ij'), -1).reshape(-1, 3)          # cs*256*256 colours
        d = (x * x).sum(1, keepdim=True) + pp[None, :] - 2 * (x.float() @ pal.float().t()).int()  # exact in f32? products < 2^24 ok
        ds, _ = torch.sort(d, dim=1)
        member = d <= ds[:, 1:2]                                                                   # T(x)
        cell = ((x[:, 1] // cs) * n + (x[:, 2] // cs)).long()
        cnt = torch.zeros(n * n, K, dtype=torch.bool, device='cuda')
        cnt.index_put_((cell,), member, accumulate=False) if False else None
        acc = torch.zeros(n * n, K, dtype=torch.int32, device='cuda')
        acc.index_add_(0, cell, member.int())
        sizes[rc * n * n:(rc + 1) * n * n] = (acc > 0).sum(1)
    s = sizes.float()
    print(f"K={K} cell {cs:2d}: mean {s.mean():.2f}  >4: {100*(s>4).float().mean():.2f}%  >5: {100*(s>5).float().mean():.2f}%  >6: {100*(s>6).float().mean():.2f}%  >8: {100*(s>8).float().mean():.2f}%  max {int(s.max())}", flush=True)
