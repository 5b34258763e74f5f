import sys; sys.path.insert(0, '.')
import torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode, ColorReducer
d = ImageDitherer(16, DitherMode.ERROR_DIFFUSION, ColorReducer.generate_uniform_palette(16), False, {"variant": "floyd_steinberg", "serpentine": "false"})
g = torch.Generator(device='cuda'); g.manual_seed(1)
for h in (64, 128, 192, 256, 512, 1024, 2160):
    f = torch.randint(0, 256, (1, h, 3840, 3), dtype=torch.uint8, device='cuda', generator=g); o = torch.empty_like(f)
    for _ in range(2): d.apply_dithering_frames(f, out=o)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); d.apply_dithering_frames(f, out=o); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"h={h:5d} bands={(h+63)//64:3d}  {min(ts):7.3f} ms", flush=True)
