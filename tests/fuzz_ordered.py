"""Randomised cross-check of the ordered / error-diffusion paths against the CPU oracle (run on the GPU box)."""
import os, sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from oracle import oracle as orc
from dither_pie_amd import backend as be
modes = ["none", "bayer", "blue_noise", "IGN", "polka_dot", "error_diffusion", "perceptual", "hybrid",
         "adaptive_variance", "ostromoukhov"]
DIFFUSERS = ("error_diffusion", "perceptual", "hybrid", "adaptive_variance", "ostromoukhov")


def run(seed, N, force=True):
  """-> number of mismatching cases.  force: per case, force a cell table / the compact kernel through the DP_* switches --
  they exist only in libditherpie_hip_exp.so (the caller selects it: _lib.select(True) / DITHER_PIE_EXPERIMENTS=1); with
  force=False the library's own choices are what is tested (the product library)."""
  orc.build()
  rs = np.random.RandomState(seed)
  bad = 0
  t0 = time.time()
  for it in range(N):
      K = int(rs.choice([2, 3, 7, 8, 9, 16, 31, 64, 100, 200, 256, 300]))
      h, w = int(rs.randint(1, 60)), int(rs.randint(1, 400))
      if rs.rand() < 0.15: h, w = int(rs.randint(100, 300)), int(rs.randint(500, 1500))
      nf = int(rs.randint(1, 4))
      gamma = bool(rs.rand() < 0.3)
      mode = modes[rs.randint(len(modes))]
      pal = orc.palr(K, seed=int(rs.randint(1 << 30)))
      if rs.rand() < 0.2: pal = pal[: K // 2] + pal[: K - K // 2]  # duplicates
      clustered = K >= 16 and rs.rand() < 0.25
      if clustered:  # as median cut produces for smooth content: most colours inside one small cube
          c0 = rs.randint(20, 200, 3); span = int(rs.choice([6, 20, 40]))
          nd = (K * 4) // 5
          pal = [tuple(int(v) for v in c0 + rs.randint(0, span, 3)) for _ in range(nd)] + pal[: K - nd]
      params = {}
      if mode == "bayer": params = {"size": str(rs.choice(["2x2", "4x4", "8x8", "16x16"]))}
      if mode == "blue_noise": params = {"size": int(rs.choice([32, 33, 40])), "seed": int(rs.randint(100))}
      if mode == "IGN": params = {"scale": float(rs.choice([1.0, 0.5, 2.5])), "seed": int(rs.randint(50))}
      if mode == "polka_dot": params = {"tile_size": int(rs.randint(4, 13)), "gamma": float(rs.choice([0.5, 1.5, 2.0]))}
      if mode == "error_diffusion":
          params = {"variant": str(rs.choice(["floyd_steinberg", "jjn", "stucki", "burkes", "atkinson", "sierra", "sierra_two_row", "sierra_lite"])),
                    "serpentine": str(rs.choice(["true", "false"]))}
          if h * w > 40000: h, w = 40, 300
      if mode == "hybrid": params = {"lum_factor": float(rs.choice([1.0, 1.4, 0.3])), "col_factor": float(rs.choice([0.2, 0.0, 1.0]))}
      if mode == "adaptive_variance": params = {"var_threshold": float(rs.choice([300.0, 50.0, 2000.0])), "window_radius": int(rs.randint(1, 4))}
      if mode == "ostromoukhov": params = {"serpentine": str(rs.choice(["true", "false"]))}
      if mode in DIFFUSERS and h * w > 40000: h, w = 40, 300
      if mode in DIFFUSERS and rs.rand() < 0.3: h, w, nf = int(rs.randint(193, 420)), int(rs.randint(64, 180)), int(rs.randint(1, 3))  # several 64-row bands: a frame spread over workgroups
      y0, x0 = (0, 0) if mode in DIFFUSERS else (int(rs.randint(0, 50)), int(rs.randint(0, 50)))
      frames = rs.randint(0, 256, (nf, h, w, 3)).astype(np.uint8)
      if clustered:  # content inside the crowded region
          frames = np.where(rs.randint(0, 3, (nf, h, w, 1)) > 0,
                            np.clip(c0 + rs.randint(-6, span + 6, (nf, h, w, 3)), 0, 255).astype(np.uint8), frames)
      if rs.rand() < 0.5:  # tie-rich content: palette colours and midpoints
          pa = np.asarray(pal, dtype=np.int64)
          pick = rs.randint(0, len(pal), (nf, h, w))
          mid = ((pa[pick] + pa[(pick + 1) % len(pal)]) // 2).astype(np.uint8)
          frames = np.where(rs.randint(0, 2, (nf, h, w, 1)) == 0, mid, frames)
      table = str(rs.choice(["", "", "u4", "u8", "w4", "w8"]))  # which cell table the accelerator uses
      # ... and, one case in three, the compact kernel on whatever 8-entry table that leaves (crowded palettes take it anyway)
      compact = rs.rand() < 0.33
      if force:
          os.environ["DP_FORCE_TABLE"] = table
          if compact: os.environ["DP_FORCE_COMPACT"] = "1"
          else: os.environ.pop("DP_FORCE_COMPACT", None)
      pal_f32, oc, lut = orc.prepare_palette(pal, gamma)
      P = be.Palette(pal_f32, oc, lut, accel=bool(rs.rand() < 0.8))
      x = torch.from_numpy(frames).cuda()
      p = dict(orc.MODE_DEFAULTS[mode]); p.update(params)
      if mode == "none": out = be.ordered(x, P, be.MODE_NEAREST, y0=y0, x0=x0)
      elif mode == "bayer": out = be.ordered(x, P, be.MODE_MATRIX, thr=be.Thresholds.from_matrix(orc.bayer_matrix(p["size"])), y0=y0, x0=x0)
      elif mode == "polka_dot": out = be.ordered(x, P, be.MODE_MATRIX, thr=be.Thresholds.from_matrix(orc.polka_dot_matrix(p["tile_size"], p["gamma"])), y0=y0, x0=x0)
      elif mode == "blue_noise": out = be.ordered(x, P, be.MODE_MATRIX, thr=be.Thresholds.blue_noise(p["size"], p["seed"]), y0=y0, x0=x0)
      elif mode == "IGN": out = be.ordered(x, P, be.MODE_IGN, ign_scale=p["scale"], ign_seed=p["seed"], y0=y0, x0=x0)
      elif mode == "error_diffusion":
          taps, div = orc.ed_kernel(p["variant"]); out = be.error_diffusion(x, P, taps, div, p["serpentine"] == "true")
      elif mode == "perceptual": out = be.variable_diffusion(x, P, be.DIFFUSER_PERCEPTUAL)
      elif mode == "hybrid": out = be.variable_diffusion(x, P, be.DIFFUSER_HYBRID, p["lum_factor"], p["col_factor"])
      elif mode == "adaptive_variance":
          out = be.variable_diffusion(x, P, be.DIFFUSER_ADAPTIVE_VARIANCE, gate=be.variance_gate(x, P, p["var_threshold"], p["window_radius"]))
      else:
          out = be.variable_diffusion(x, P, be.DIFFUSER_OSTROMOUKHOV, serpentine=(p["serpentine"] == "true"),
                                      coef=torch.from_numpy(orc.ostromoukhov_coefficients()).cuda())
      out = out.cpu().numpy()
      for i in range(nf):
          ref = orc.apply_dithering(frames[i], pal, mode, params, gamma, y0=y0, x0=x0) if mode not in DIFFUSERS else orc.apply_dithering(frames[i], pal, mode, params, gamma)
          if not np.array_equal(out[i], ref):
              bad += 1
              print("MISMATCH", it, mode, params, "K", K, "gamma", gamma, (nf, h, w), "y0x0", (y0, x0), "accel", P.accel_entries, int((out[i] != ref).any(-1).sum()), "px", flush=True)
              break
  os.environ.pop("DP_FORCE_TABLE", None)
  os.environ.pop("DP_FORCE_COMPACT", None)
  print(f"fuzz: {N} cases, {bad} mismatching, {time.time()-t0:.1f} s")
  return bad


if __name__ == "__main__":
    from dither_pie_amd import _lib
    if not _lib.EXPERIMENTS: _lib.select(True)
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 300) else 0)
