#!/bin/bash
# PMC passes for the dominant kernel (one rocprofv3 run per counter group; never combined with tracing)
# usage: profiles/pmc_pass.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc}
export TMPDIR=/tmp
mkdir -p "$OUT"
CMD="python3 bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/p1" -- $CMD > "$OUT/p1.log" 2>&1 || exit 11
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/p2" -- $CMD > "$OUT/p2.log" 2>&1 || exit 12
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/p3" -- $CMD > "$OUT/p3.log" 2>&1 || exit 13
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p4" -- $CMD > "$OUT/p4.log" 2>&1 || exit 14
echo pmc_done
