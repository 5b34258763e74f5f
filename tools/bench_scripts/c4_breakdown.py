"""Where C4's time goes: seeding sample, k-means++ on the host, the Lloyd loop on the device, the blue-noise dither."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import kmeans, backend as be
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
g = torch.Generator(device='cuda'); g.manual_seed(99)
img = torch.randint(0, 256, (4320, 7680, 3), dtype=torch.uint8, device='cuda', generator=g)
px = img.reshape(-1, 3)
def T(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
t, sample = T(lambda: kmeans.seed_sample(px, px.shape[0], 0, 42)); print(f"seed_sample      {t:8.2f} ms")
t, init = T(lambda: kmeans.kmeans_plusplus(sample, 32, np.random.RandomState(42))); print(f"kmeans_plusplus  {t:8.2f} ms (host)")
t, res = T(lambda: kmeans.lloyd(px, init)); print(f"lloyd            {t:8.2f} ms  ({res[2]} iterations)")
pal = [tuple(int(v) for v in c) for c in res[0].astype(int)]
d = ImageDitherer(32, DitherMode.BLUE_NOISE, pal, False, {"size": 64, "seed": 42})
t, _ = T(lambda: d.apply_dithering_frames(img)); print(f"blue-noise dither{t:8.2f} ms")
t, _ = T(lambda: kmeans.fit_palette(px, 32, 42)); print(f"fit_palette      {t:8.2f} ms")
c = torch.from_numpy(init).cuda(); tot = torch.zeros(160, dtype=torch.int64, device='cuda')
t, _ = T(lambda: be.kmeans_step_into(px, c, tot, want_sq=False), 20); print(f"one pass         {t:8.3f} ms")
