"""Lloyd over the colour histogram against Lloyd over the pixels on the C4 image (33 M pixels), event-timed on the product
library: histogram build, one pass of either kind (memset and list build included), totals compared; then the whole fit
(k-means++ seeding + Lloyd) both ways.  Contents: uniform noise, smooth image-like content with grain, a flat frame.
usage: kmeans_hist_time.py [K ...]"""
import sys; sys.path.insert(0, '.')
import time
import numpy as np, torch
from dither_pie_amd import backend as be, kmeans


def ev_time(fn, reps=6):
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


g = torch.Generator(device='cuda'); g.manual_seed(99)
N = 4320 * 7680
rnd = torch.randint(0, 256, (N, 3), dtype=torch.uint8, device='cuda', generator=g)
yy, xx = torch.meshgrid(torch.arange(4320, device='cuda'), torch.arange(7680, device='cuda'), indexing='ij')
smooth = torch.stack([(xx * 255 // 7679), (yy * 255 // 4319), ((xx + yy) * 255 // (7679 + 4319))], -1).to(torch.uint8).reshape(-1, 3)
smooth = (smooth.to(torch.int16) + torch.randint(-6, 7, smooth.shape, device='cuda', generator=g).to(torch.int16)).clamp(0, 255).to(torch.uint8).contiguous()
flat = torch.tensor([17, 99, 200], dtype=torch.uint8, device='cuda').repeat(N, 1).contiguous()
del yy, xx
for name, px in (("uniform random", rnd), ("smooth + grain", smooth), ("flat", flat)):
    hist = be.ColourHistogram(px)
    t_build = ev_time(lambda: hist.add(px, accumulate=False), 4)
    occ = int(hist.buf[1 << 26:].view(torch.int32)[4096].item())
    distinct = int((hist.buf[:1 << 26].view(torch.int32) != 0).sum().item())
    print(f"{name:15s} histogram build {t_build:.4f} ms ({N / t_build * 1e-6:.1f} Gpx/s), {distinct} distinct colours in {occ} occupied cells", flush=True)
    for K in ([int(a) for a in sys.argv[1:]] or [1, 8, 32, 128, 256]):
        c = torch.from_numpy(np.random.RandomState(1).rand(K, 3) * 255.0).cuda()
        for _ in range(3):   # a few Lloyd steps so that the centres look like k-means centres of this data
            s, n_, _q = hist.step(c)
            c = torch.where(n_[:, None] > 0, s.double() / n_.clamp(min=1)[:, None].double(), c).contiguous()
        tot_p = torch.zeros(5 * K, dtype=torch.int64, device='cuda'); tot_h = torch.zeros_like(tot_p)
        be.kmeans_step_into(px, c, tot_p, want_sq=True); hist.step_into(c, tot_h, True)
        same = bool(torch.equal(tot_p, tot_h))
        tp = ev_time(lambda: be.kmeans_step_into(px, c, tot_p, want_sq=False))
        th = ev_time(lambda: hist.step_into(c, tot_h, False))
        print(f"{name:15s} K={K:3d}: pass over pixels {tp:.4f} ms   over the histogram {th:.4f} ms   totals equal: {same}", flush=True)
    if name != "flat":
        for K in (32,):
            sample = kmeans.seed_sample(px, N, 0, 42, None, as_tensor=True)
            init = kmeans.kmeans_plusplus_device(sample, K, np.random.RandomState(42))
            res = {}
            for h in (False, True):
                kmeans.lloyd(px, init, histogram=h)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                res[h] = kmeans.lloyd(px, init, histogram=h)
                torch.cuda.synchronize(); res[h] = res[h] + (time.perf_counter() - t0,)
            same = bool(np.array_equal(res[False][0], res[True][0])) and res[False][2] == res[True][2]
            print(f"{name:15s} K={K}: whole Lloyd fit, {res[True][2]} iterations: over pixels {res[False][3] * 1e3:.2f} ms, over the histogram "
                  f"{res[True][3] * 1e3:.2f} ms (build included); same centres: {same}", flush=True)
