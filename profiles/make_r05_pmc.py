#!/usr/bin/env python3
"""Assemble profiles/r05_pmc_legs.json from pmc_summarise.py outputs of profiles/pmc_pass_cmd4.sh runs (round 5): for every
non-headline bench leg the SQ counters AND the HBM traffic (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, both reported in KiB;
MI355X_MICROARCH.md, HBM) per launch of its dominant kernel, next to the leg's algorithmic bytes.

usage: profiles/make_r05_pmc.py <dir with pmc_<leg>_summary.json files>
Stamped with the sha256 of the sources each kernel was built from; bench.py reports a leg's `traffic` only while that stamp
equals the tree's."""
import hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PX24 = 24 * 2160 * 3840
N8K = 4320 * 7680


def sha16(names):
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(ROOT, "dither_pie_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


ORDERED_SRC = ("ordered.hip", "accel.hip", "dp_internal.h", "tree_query.hip.h")
KMEANS_SRC = ("kmeans_hist.hip", "kmeans_label.hip.h", "wave_util.hip.h")
ED_SRC = ("ediff.hip", "ed_nearest.hip.h", "dp_internal.h")

# leg -> (summary file, kernel substring, algorithmic bytes per launch, unit count for per-unit figures, sources, workload)
LEGS = {
    "c2_crowded": ("pmc_crowded_summary.json", "ordered_compact_kernel", 6 * PX24, PX24, ORDERED_SRC,
                   "tools/bench_scripts/crowded_prof.py: 24 image-like 4K frames + their own median-cut 256 palette, Bayer 8x8"),
    "c2_use_gamma": ("pmc_gamma_summary.json", "ordered_compact_float_kernel", 6 * PX24, PX24, ORDERED_SRC,
                     "tools/bench_scripts/gamma_prof.py: C2 with use_gamma=True (palr(256), Bayer 8x8), 24 4K frames"),
    "c4_kmeans_pass": ("pmc_khist_noise_summary.json", "hist_pass_kernel<false, false>", 4096 * 16384, N8K, KMEANS_SRC,
                       "tools/bench_scripts/prof_kmeans_hist.py 32 noise: one Lloyd pass over the histogram of 33 M noise pixels "
                       "(all 4096 cells occupied: 64 MB)"),
    "c4_kmeans_pass_image_like": ("pmc_khist_smooth_summary.json", "hist_pass_kernel<false, false>", None, N8K, KMEANS_SRC,
                                  "tools/bench_scripts/prof_kmeans_hist.py 32 smooth: the same on image-like content (852 occupied cells)"),
    "c5_video": ("pmc_c5_summary.json", "ordered_lean_kernel<1, 4", 6 * 100 * 1080 * 1920, 100 * 1080 * 1920, ORDERED_SRC,
                 "tools/bench_scripts/c5_prof.py: C5's launch -- 100 1080p noise frames, Bayer 4x4, 16 uniform colours"),
    "c3": ("pmc_ed_summary.json", "ed_wavefront_kernel", 6 * 256 * 2160 * 3840, 256 * 2160 * 3840, ED_SRC,
           "tools/bench_scripts/ed_prof.py 16 256: Floyd-Steinberg, 16 uniform colours, 256 4K frames in flight -- the launch shape of "
           "bench.py's c3 leg itself (round 4 took the ratio on 64 frames and applied it to 256)"),
}
BUILD_KERNELS = ("hist_count_kernel", "hist_plan_kernel", "hist_scatter_kernel", "hist_parts_kernel")


def derive(c, alg_bytes, units):
    d = {}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd, wr = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024
        d.update(hbm_read_bytes_FETCH_SIZE_x2_gfx950=rd, hbm_write_bytes_WRITE_SIZE=wr, hbm_traffic_bytes_per_launch=rd + wr)
        if alg_bytes:
            d.update(algorithmic_bytes_per_launch=alg_bytes, traffic_over_algorithmic=(rd + wr) / alg_bytes)
    if "SQ_INSTS_VALU" in c:
        d["valu_wave_instructions_per_launch"] = c["SQ_INSTS_VALU"]
        d["valu_wave_instructions_per_unit"] = c["SQ_INSTS_VALU"] * 64 / units
        d["valu_issue_time_ms_at_4p3_cycles_2p4GHz"] = c["SQ_INSTS_VALU"] * 4.3 / (1024 * 2.4e9) * 1e3
    if "SQ_INSTS_SALU" in c:
        d["salu_per_256_units"] = c["SQ_INSTS_SALU"] * 256 / units
    if "SQ_INSTS_LDS" in c:
        d["lds_instructions_per_unit"] = c["SQ_INSTS_LDS"] * 64 / units
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_fraction_of_lds_cycles"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES"):
        d["wait_any_fraction_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        d["wait_inst_any_fraction_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    return d


def main():
    root = sys.argv[1]
    out = {"note": "SQ_* cycle counters are in quad-cycles; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled as MI355X_MICROARCH.md "
                   "prescribes for gfx950; one rocprofv3 --pmc run per counter group (profiles/pmc_pass_cmd4.sh), never combined with "
                   "tracing; 'units' are pixels (dither legs, histogram build) or pixels of the image behind the histogram (Lloyd pass)",
           "legs": {}}
    for leg, (fn, sub, alg, units, src, workload) in LEGS.items():
        path = os.path.join(root, fn)
        if not os.path.exists(path):
            continue
        j = json.load(open(path))["counters_mean_per_launch"]
        name = next((k for k in j if sub in k), None)
        if name is None:
            continue
        out["legs"][leg] = {"kernel": name, "workload": workload, "counters_mean_per_launch": j[name], "derived": derive(j[name], alg, units),
                            "kernel_sources_sha16": sha16(src)}
    # the histogram build is four kernels: their counters side by side, traffic summed
    for kind in ("noise", "smooth"):
        path = os.path.join(root, f"pmc_khist_{kind}_summary.json")
        if not os.path.exists(path):
            continue
        j = json.load(open(path))["counters_mean_per_launch"]
        parts, traffic = {}, 0.0
        for k, c in j.items():
            if any(b in k for b in BUILD_KERNELS):
                parts[k] = {"counters_mean_per_launch": c, "derived": derive(c, None, N8K)}
                traffic += parts[k]["derived"].get("hbm_traffic_bytes_per_launch", 0.0)
        if parts:
            out["legs"]["c4_kmeans_histogram" + ("" if kind == "noise" else "_image_like")] = {
                "kernels": parts, "workload": f"tools/bench_scripts/prof_kmeans_hist.py 32 {kind}: pixels -> count[colour], 33 M pixels",
                "derived": {"hbm_traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": 3 * N8K,
                            "traffic_over_algorithmic": traffic / (3 * N8K)},
                "kernel_sources_sha16": sha16(KMEANS_SRC)}
    # r05_pmc_kmeans.json: the Lloyd pass before / after on one box -- over the pixels (round 3's kmeans_cells_kernel + its list build)
    # against over the colour histogram (hist_pass_kernel), and what the histogram costs to build
    kp = os.path.join(root, "pmc_kpix_summary.json")
    if os.path.exists(kp) and "c4_kmeans_pass" in out["legs"]:
        j = json.load(open(kp))["counters_mean_per_launch"]
        before = {k: {"counters_mean_per_launch": c, "derived": derive(c, 3 * N8K if "cells_kernel" in k else None, N8K)}
                  for k, c in j.items() if "kmeans_cells" in k}
        km = {"workload": "K = 32 over the 33 M pixels of a 7680x4320 noise image (tools/bench_scripts/prof_kmeans.py / prof_kmeans_hist.py), same box, same session",
              "before_pass_over_the_pixels": before,
              "after_pass_over_the_histogram": out["legs"]["c4_kmeans_pass"],
              "after_pass_over_the_histogram_image_like": out["legs"].get("c4_kmeans_pass_image_like"),
              "histogram_build_once_per_fit": out["legs"].get("c4_kmeans_histogram"),
              "histogram_build_once_per_fit_image_like": out["legs"].get("c4_kmeans_histogram_image_like"),
              "note": out["note"], "kernel_sources_sha16": sha16(KMEANS_SRC)}
        json.dump(km, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "r05_pmc_kmeans.json"), "w"), indent=1, sort_keys=True)
        for k, v in before.items():
            d = v["derived"]
            print(f"before: {k[:60]:60s} VALU/px {d.get('valu_wave_instructions_per_unit', 0):.1f} traffic {d.get('hbm_traffic_bytes_per_launch', 0) / 1e6:.1f} MB")
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "r05_pmc_legs.json")
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    for leg, o in out["legs"].items():
        d = o["derived"]
        print(f"{leg:32s} traffic {d.get('hbm_traffic_bytes_per_launch', 0) / 1e6:9.1f} MB  x{d.get('traffic_over_algorithmic', 0):.2f} of algorithmic"
              + (f"  VALU/unit {d['valu_wave_instructions_per_unit']:.1f}" if 'valu_wave_instructions_per_unit' in d else ""))


if __name__ == "__main__":
    main()
