#!/bin/bash
# SQ counters of the bench workload, two passes (8 SQ counters each).  usage: scripts/pmc_sq.sh <tag> [bench args]
set -u
TAG=${1:-sq}; shift || true
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-graph $*"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d "$OUT/p1" -- python "$REPO/bench.py" $ARGS > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 \
  --output-format csv -d "$OUT/p2" -- python "$REPO/bench.py" $ARGS > "$OUT/p2.log" 2>&1
cd "$REPO"
python scripts/pmc_summary.py $(find "$OUT" -name "*counter_collection.csv") | tee "$OUT/summary.txt"
