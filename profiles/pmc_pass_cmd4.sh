#!/bin/bash
# The four counter passes of MI355X_MICROARCH.md (HBM / rocprofv3) for an arbitrary command: two SQ groups, FETCH_SIZE, WRITE_SIZE
# -- one rocprofv3 run per group, never combined with tracing.
# usage: profiles/pmc_pass_cmd4.sh <outdir> <program> [args...]      (the program itself, no env/sh wrappers)
set -u
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/p1" -- "$@" > "$OUT/p1.log" 2>&1 || exit 11
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/p2" -- "$@" > "$OUT/p2.log" 2>&1 || exit 12
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/p3" -- "$@" > "$OUT/p3.log" 2>&1 || exit 13
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p4" -- "$@" > "$OUT/p4.log" 2>&1 || exit 14
echo pmc_done
