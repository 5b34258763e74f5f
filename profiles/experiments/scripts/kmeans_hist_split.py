"""hist_pass_kernel with a cell over 1, 2 or 4 waves (DP_KMEANS_HIST_SPLIT, experiment library) on the C4 noise image and on
image-like content, K = 8 / 32 / 256; event-timed pass incl. its memset."""
import os; os.environ["DITHER_PIE_EXPERIMENTS"] = "1"
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import backend as be
g = torch.Generator(device='cuda'); g.manual_seed(99)
N = 4320 * 7680
rnd = torch.randint(0, 256, (N, 3), dtype=torch.uint8, device='cuda', generator=g)
yy, xx = torch.meshgrid(torch.arange(4320, device='cuda'), torch.arange(7680, device='cuda'), indexing='ij')
sm = torch.stack([(xx * 255 // 7679), (yy * 255 // 4319), ((xx + yy) * 255 // (7679 + 4319))], -1).to(torch.int16).reshape(-1, 3)
sm = (sm + torch.randint(-6, 7, sm.shape, device='cuda', generator=g).to(torch.int16)).clamp(0, 255).to(torch.uint8).contiguous()
for name, px in (("noise", rnd), ("smooth", sm)):
    hist = be.ColourHistogram(px)
    for K in (8, 32, 256):
        c = torch.from_numpy(np.random.RandomState(1).rand(K, 3) * 255.0).cuda()
        for _ in range(3):
            s_, n_, _q = hist.step(c)
            c = torch.where(n_[:, None] > 0, s_.double() / n_.clamp(min=1)[:, None].double(), c).contiguous()
        tot = torch.zeros(5 * K, dtype=torch.int64, device='cuda')
        res = []
        for split in ("0", "1", "2", "4"):
            os.environ["DP_KMEANS_HIST_SPLIT"] = split
            for _ in range(2): hist.step_into(c, tot, False)
            ts = []
            for _ in range(8):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); hist.step_into(c, tot, False); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            res.append(f"split {split}: {min(ts)*1e3:.1f} us")
        print(name, "K", K, " | ".join(res), flush=True)
