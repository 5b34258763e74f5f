#!/bin/bash
# PMC passes for the error-diffusion wavefront kernel (one rocprofv3 run per counter group; never combined with tracing)
# usage: profiles/pmc_pass_ed.sh <outdir>     (64 4K frames, Floyd-Steinberg, 16 colours, one workgroup per frame)
set -u
OUT=${1:-gpurun_out/pmc_ed}
export TMPDIR=/tmp
export DP_ED_ONE_WG=1
mkdir -p "$OUT"
CMD="python3 tools/bench_scripts/ed_prof.py"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/p1" -- $CMD > "$OUT/p1.log" 2>&1 || exit 11
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/p2" -- $CMD > "$OUT/p2.log" 2>&1 || exit 12
echo pmc_ed_done
