import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/dither_pie_amd") else ".")
os.environ["DITHER_PIE_EXPERIMENTS"]="1"; os.environ["DP_DEBUG_ACCEL"]="1"
import numpy as np, torch
from dither_pie_amd.dithering_lib import ImageDitherer, DitherMode
g=torch.Generator(device="cuda"); g.manual_seed(1)
frames=torch.randint(0,256,(24,2160,3840,3),dtype=torch.uint8,device="cuda",generator=g); out=torch.empty_like(frames)
for K in (300, 512, 768, 1024):
    pal=[tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0,256,(K,3))]
    d=ImageDitherer(K, DitherMode.BAYER, pal, False, {"size":"8x8"}).prepare()
    d.apply_dithering_frames(frames,out=out); torch.cuda.synchronize()
    ts=[]
    for _ in range(3):
        t=time.perf_counter(); d.apply_dithering_frames(frames,out=out); torch.cuda.synchronize(); ts.append((time.perf_counter()-t)*1e3)
    print("K",K,"ms per 24 frames",round(min(ts),2),flush=True)
