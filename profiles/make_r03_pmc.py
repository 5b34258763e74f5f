#!/usr/bin/env python3
"""Assemble profiles/r03_pmc_crowded.json and r03_pmc_float.json from pmc_summarise.py outputs (before / after on the same box).

usage: profiles/make_r03_pmc.py <dir with pmc_crowded_summary.json, pmc_crowded_before_summary.json, pmc_gamma_summary.json,
                                 pmc_gamma_before_summary.json> [first-version summary json]
Stamped with the sha256 of the kernel sources like r03_pmc_ordered.json (pmc_derive.py)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_derive import kernel_sources_sha16

PX = 24 * 2160 * 3840


def derived(c):
    d = {"valu_wave_instructions_per_pixel": c["SQ_INSTS_VALU"] * 64 / PX, "salu_per_256px_wave_tile": c["SQ_INSTS_SALU"] * 256 / PX,
         "lds_instructions_per_pixel": c["SQ_INSTS_LDS"] * 64 / PX}
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_fraction_of_lds_cycles"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "SQ_INSTS_VMEM_RD" in c:
        d["vector_loads_per_launch"] = c["SQ_INSTS_VMEM_RD"]
    if "SQ_WAIT_ANY" in c:
        d["wait_any_fraction_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        d["wait_inst_any_fraction_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    d["valu_issue_time_ms_at_4p3_cycles_2p4GHz"] = c["SQ_INSTS_VALU"] * 4.3 / (1024 * 2.4e9) * 1e3
    return d


def main_kernel(path, sub):
    j = json.load(open(path))
    name = next(k for k in j["counters_mean_per_launch"] if sub in k)
    c = j["counters_mean_per_launch"][name]
    return {"kernel": name, "counters_mean_per_launch": c, "derived": derived(c)}


def main():
    root = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else None
    stamp = kernel_sources_sha16()
    note = ("SQ_* cycle counters are in quad-cycles; one rocprofv3 --pmc run per counter group (profiles/pmc_pass_cmd.sh), never combined "
            "with tracing; 24 frames of 3840x2160 per launch; 'before' = the round-2 kernel on the same box in the same session "
            "(DITHER_PIE_EXPERIMENTS=1 DP_NO_COMPACT_KERNEL=1).")
    crowded = {"workload": "tools/bench_scripts/crowded_prof.py: image-like frames + their own median-cut 256 palette, Bayer 8x8",
               "before_round2_kernel": main_kernel(os.path.join(root, "pmc_crowded_before_summary.json"), "ordered_lean_kernel"),
               "after": main_kernel(os.path.join(root, "pmc_crowded_summary.json"), "ordered_compact_kernel"),
               "kernel_sources_sha16": stamp, "pixels_per_launch": PX, "note": note}
    if first and os.path.exists(first):
        crowded["first_version_of_the_compact_kernel"] = dict(main_kernel(first, "ordered_compact_kernel"),
                                                              what="4-byte colours gathered through the index bytes + the lean kernel's cand8 on positions, "
                                                                   "tie codes for every tie, compare chain for the pick, per-pixel flag atomics: 1.131 ms")
    json.dump(crowded, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "r03_pmc_crowded.json"), "w"), indent=1, sort_keys=True)
    flt = {"workload": "tools/bench_scripts/gamma_prof.py: C2 with use_gamma=True (palr(256), Bayer 8x8)",
           "before_round2_kernel": main_kernel(os.path.join(root, "pmc_gamma_before_summary.json"), "ordered_lean_float_kernel"),
           "after": main_kernel(os.path.join(root, "pmc_gamma_summary.json"), "ordered_compact_float_kernel"),
           "kernel_sources_sha16": stamp, "pixels_per_launch": PX, "note": note}
    json.dump(flt, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "r03_pmc_float.json"), "w"), indent=1, sort_keys=True)
    for nm, o in (("crowded", crowded), ("float", flt)):
        for k in ("before_round2_kernel", "first_version_of_the_compact_kernel", "after"):
            if k in o:
                d = o[k]["derived"]
                print(f"{nm:8s} {k:36s} VALU/px {d['valu_wave_instructions_per_pixel']:.1f}  LDS/px {d['lds_instructions_per_pixel']:.1f}  "
                      f"loads {d.get('vector_loads_per_launch', 0)/1e6:.2f} M  wait_any {d.get('wait_any_fraction_of_wave_cycles', 0):.2f}")


if __name__ == "__main__":
    main()
