"""Soak of the one-launch Lloyd iteration (dp_kmeans_hist_iterate: every workgroup adds its totals with atomics, the workgroup
that draws the last ticket runs the centre update) against the three-step iteration (pass / update as separate launches) and the
pass over the pixels: exact equality of centres, inertia and iteration counts, fit after fit -- an ordering bug between the totals
and the ticket would show up as a fit that differs.  usage: fuzz_kmeans_fused.py [rounds]   (run on the GPU box)"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from dither_pie_amd import kmeans


def run(rounds=40):
    rs = np.random.RandomState(77)
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    n = 3_000_000
    noise = torch.randint(0, 256, (n, 3), dtype=torch.uint8, device='cuda', generator=g)
    yy, xx = torch.meshgrid(torch.arange(1500, device='cuda'), torch.arange(2000, device='cuda'), indexing='ij')
    smooth = torch.stack([(xx * 255 // 1999), (yy * 255 // 1499), ((xx + yy) * 255 // 3498)], -1).to(torch.int16).reshape(-1, 3)
    smooth = (smooth + torch.randint(-6, 7, smooth.shape, device='cuda', generator=g).to(torch.int16)).clamp(0, 255).to(torch.uint8).contiguous()
    bad = fits = 0
    t0 = time.time()
    for r in range(rounds):
        for name, px in (("noise", noise), ("smooth", smooth)):
            K = int(rs.choice([2, 8, 32, 100, 256]))
            host = px[torch.from_numpy(rs.randint(0, px.shape[0], K)).cuda()].cpu().numpy().astype(np.float64)
            init = np.clip(host + rs.choice([0.0, 0.25, 0.5]), 0.0, 255.0)
            a = kmeans.lloyd(px, init, max_iter=25, histogram=True)               # fused
            b = kmeans.lloyd(px, init, max_iter=25, histogram=True, fuse=False)   # pass + update
            fits += 1
            if not (np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2]):
                bad += 1
                print("MISMATCH fused vs three-step", r, name, K, a[2], b[2], float(np.abs(a[0] - b[0]).max()), flush=True)
            if r % 8 == 0:
                c = kmeans.lloyd(px, init, max_iter=25, histogram=False)
                if not (np.array_equal(a[0], c[0]) and a[1] == c[1] and a[2] == c[2]):
                    bad += 1
                    print("MISMATCH histogram vs pixels", r, name, K, flush=True)
    print(f"fuzz_kmeans_fused: {fits} fits of up to 25 iterations each way, {bad} mismatching, {time.time() - t0:.1f} s")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40) else 0)
