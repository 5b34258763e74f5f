"""Probe: the fused Lloyd iteration (dp_kmeans_hist_iterate) launched 16 times eagerly against a captured graph of the same
16 launches -- capture + instantiate cost, replay time.  usage: kmeans_graph_probe.py [noise|smooth]"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import backend
kind = sys.argv[1] if len(sys.argv) > 1 else "smooth"
h, w, K = 4320, 7680, 32
if kind == "noise":
    img = torch.from_numpy(np.random.RandomState(99).randint(0, 256, (h, w, 3), dtype=np.uint8)).cuda()
else:
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([80 + 60 * np.sin(x / 300.0) + 40 * (y / h), 110 + 50 * np.cos(y / 200.0) + 20 * np.sin(x / 97.0), 160 + 70 * (y / h) + 10 * np.sin((x + y) / 50.0)], -1)
    img = torch.from_numpy(np.clip(a + np.random.RandomState(3).normal(0, 3, a.shape), 0, 255).astype(np.uint8)).cuda()
hist = backend.ColourHistogram(img.reshape(-1, 3))
dev = img.device
c0 = torch.as_tensor(np.random.RandomState(1).uniform(20, 235, (K, 3))).to(dev).contiguous()
def state():
    return (c0.clone(), torch.zeros(5 * K, dtype=torch.int64, device=dev), torch.zeros(4 * K, dtype=torch.int64, device=dev),
            torch.zeros(8, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
def sync_time(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
c, tot, prev, st, tk = state()
hist.iterate(c, tot, prev, st, tk, 0.0, 10000, True)
for rep in range(3):
    print("eager, 16 iterations: %.3f ms" % sync_time(lambda: [hist.iterate(c, tot, prev, st, tk, 0.0, 10000, False) for _ in range(16)]), flush=True)
g = torch.cuda.CUDAGraph()
def capture():
    with torch.cuda.graph(g):
        for _ in range(16): hist.iterate(c, tot, prev, st, tk, 0.0, 10000, False)
print("capture + instantiate: %.3f ms" % sync_time(capture), flush=True)
for rep in range(3):
    print("graph replay, 16 iterations: %.3f ms" % sync_time(g.replay), flush=True)
