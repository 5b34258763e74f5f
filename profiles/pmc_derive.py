#!/usr/bin/env python3
"""Turn a pmc_summarise.py result into the file bench.py reads (profiles/r02_pmc_ordered.json).

usage: profiles/pmc_derive.py <summary.json> <out.json> <kernel-substring> <pixels per launch> [kernel ms]
Adds the derived figures (per-launch HBM traffic = FETCH_SIZE x 2 (gfx950: MI355X_MICROARCH.md, HBM) + WRITE_SIZE, both
reported in KiB; VALU instructions per pixel; LDS bank-conflict share) and stamps the file with the hash of the kernel
sources it was taken with (bench.py ignores the file when the sources have changed since)."""
import hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha16():
    hsh = hashlib.sha256()
    for name in ("ordered.hip", "accel.hip", "dp_internal.h", "tree_query.hip.h"):
        path = os.path.join(ROOT, "dither_pie_amd", "csrc", name)
        if os.path.exists(path):
            with open(path, "rb") as f:
                hsh.update(f.read())
    return hsh.hexdigest()[:16]


def main():
    src, out, sub, px = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    kms = float(sys.argv[5]) if len(sys.argv) > 5 else None
    j = json.load(open(src))
    name = next(k for k in j["counters_mean_per_launch"] if sub in k)
    c = j["counters_mean_per_launch"][name]
    d = {}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd, wr = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024
        d.update(hbm_read_bytes_FETCH_SIZE_x2_gfx950=rd, hbm_write_bytes_WRITE_SIZE=wr, hbm_traffic_bytes_per_launch=rd + wr,
                 algorithmic_bytes_per_launch=6 * px, traffic_over_algorithmic=(rd + wr) / (6 * px))
    if "SQ_INSTS_VALU" in c:
        d["valu_wave_instructions_per_launch"] = c["SQ_INSTS_VALU"]
        d["valu_wave_instructions_per_pixel"] = c["SQ_INSTS_VALU"] * 64 / px
    if "SQ_INSTS_SALU" in c:
        d["salu_per_256px_wave_tile"] = c["SQ_INSTS_SALU"] * 256 / px
    if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
        d["lds_bank_conflict_fraction_of_lds_cycles"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if kms:
        d["kernel_ms_from_bench_hip_events"] = kms
        if "SQ_INSTS_VALU" in c:
            d["valu_issue_time_ms_at_4p3_cycles_2p4GHz"] = c["SQ_INSTS_VALU"] * 4.3 / (1024 * 2.4e9) * 1e3
    j.update(kernel=name, pixels_per_launch=px, derived=d, kernel_sources_sha16=kernel_sources_sha16(),
             note="FETCH_SIZE/WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for "
                  "gfx950. SQ_* cycle counters are in quad-cycles. One rocprofv3 --pmc run per counter group "
                  "(profiles/pmc_pass.sh), never combined with tracing.")
    json.dump(j, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(d, indent=1))


if __name__ == "__main__":
    main()
