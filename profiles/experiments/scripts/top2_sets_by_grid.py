"""Size of the exact top-2 candidate set T(cell) -- the union over the cell's integer colours of everything at most as far as
the second nearest entry -- for the headline palette on grids coarser than 16^3 (8-byte candidate
records {rgb, |p|^2 << 8 | offset} need a 12^3..14^3 grid to fit 160 KB).  Runs on the GPU (torch), ~seconds."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pal = torch.tensor(np.random.RandomState(7).randint(0, 256, (K, 3)), dtype=torch.int32, device='cuda')
pp = (pal * pal).sum(1)
ax = torch.arange(256, device='cuda', dtype=torch.int32)
for cs in (16, 19, 20, 22):
    n = (255 // cs) + 1
    sizes = torch.zeros(n * n * n, dtype=torch.int32, device='cuda')
    for rc in range(n):
        r = ax[rc * cs:min(256, (rc + 1) * cs)]
        x = torch.stack(torch.meshgrid(r, ax, ax, indexing='ij'), -1).reshape(-1, 3)
        d = (x * x).sum(1, keepdim=True) + pp[None, :] - 2 * (x.float() @ pal.float().t()).int()
        ds, _ = torch.sort(d, dim=1)
        member = d <= ds[:, 1:2]
        cell = ((x[:, 1] // cs) * n + (x[:, 2] // cs)).long()
        acc = torch.zeros(n * n, K, dtype=torch.int32, device='cuda')
        acc.index_add_(0, cell, member.int())
        sizes[rc * n * n:(rc + 1) * n * n] = (acc > 0).sum(1)
    s = sizes.float()
    print(f"cell width {cs}: {n}^3 = {n**3} cells, {n**3 * 64 // 1024} KB as 8 records of 8 bytes; |T| mean {s.mean():.2f}, "
          f"> 8 in {100*(s>8).float().mean():.1f} % of the cells, max {int(s.max())}", flush=True)
