"""Where C4's time goes (round 4: the fit runs over the colour histogram): seeding sample, k-means++ on the device, histogram
build, the Lloyd loop, the palette / threshold objects and the blue-noise dither -- wall-clock of each step, synchronised."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import kmeans, backend as be
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
img = torch.from_numpy(np.random.RandomState(99).randint(0, 256, (4320, 7680, 3), dtype=np.uint8)).cuda()
px = img.reshape(-1, 3)
def T(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
t, sample = T(lambda: kmeans.seed_sample(px, px.shape[0], 0, 42, None, as_tensor=True)); print(f"seed_sample           {t:8.3f} ms")
t, init = T(lambda: kmeans.kmeans_plusplus_device(sample, 32, np.random.RandomState(42))); print(f"kmeans_plusplus_device{t:8.3f} ms")
t, hist = T(lambda: be.ColourHistogram(px)); print(f"ColourHistogram(px)   {t:8.3f} ms")
t, res = T(lambda: kmeans.lloyd(px, init)); print(f"lloyd                 {t:8.3f} ms  ({res[2]} iterations, histogram build included)")
pal = [tuple(int(v) for v in c) for c in res[0].astype(int)]
t, d = T(lambda: ImageDitherer(32, DitherMode.BLUE_NOISE, pal, False, {"size": 64, "seed": 42})); print(f"ImageDitherer(...)    {t:8.3f} ms")
t, _ = T(lambda: d.apply_dithering_frames(img)); print(f"blue-noise dither     {t:8.3f} ms")
t, _ = T(lambda: kmeans.fit_palette(px, 32, 42)); print(f"fit_palette           {t:8.3f} ms")
def c4():
    p, _, _, it = kmeans.fit_palette(px, 32, 42)
    ImageDitherer(32, DitherMode.BLUE_NOISE, p, False, {"size": 64, "seed": 42}).apply_dithering_frames(img)
t, _ = T(c4); print(f"C4 (fit + dither)     {t:8.3f} ms")
