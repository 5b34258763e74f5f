"""C5's launch (100 x 1080p noise frames, Bayer 4x4, 16 uniform colours, prepared palette): a few launches for rocprofv3 passes.
usage: python3 tools/bench_scripts/c5_prof.py [launches]"""
import sys; sys.path.insert(0, '.')
import torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
g = torch.Generator(device='cuda'); g.manual_seed(1234)
f = torch.randint(0, 256, (100, 1080, 1920, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
d = ImageDitherer(16, DitherMode.BAYER, ColorReducer.generate_uniform_palette(16), False, {"size": "4x4"}).prepare()
for _ in range(3): d.apply_dithering_frames(f, out=o)
ts = []
for _ in range(n):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); d.apply_dithering_frames(f, out=o); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(f"C5 launch, 100 x 1080p: min {min(ts):.3f} ms = {100 / min(ts) * 1e3:.0f} frames/s", flush=True)
