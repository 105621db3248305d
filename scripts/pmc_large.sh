#!/bin/bash
# SQ + GRBM counters of one bench workload, two passes.  usage (GPU box): scripts/pmc_large.sh <tag> <workload> [bench args]
#   -> gpurun_out/pmc_<tag>_<workload>/summary.md
set -u
TAG=$1; W=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_${TAG}_$W; mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS="--workload $W --steps 12 --warmup 3 --repeats 1 --no-cpu-baseline --no-secondary --no-graph $*"
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS \
  --output-format csv -d "$OUT/p1" -- python "$REPO/bench.py" $ARGS > "$OUT/p1.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU \
  --output-format csv -d "$OUT/p2" -- python "$REPO/bench.py" $ARGS > "$OUT/p2.log" 2>&1
cd "$REPO"
{ echo "# SQ / GRBM counters, bench.py --workload $W (eager launches, 12 steps; rocprofv3 --pmc, two passes)"; echo;
  python scripts/pmc_summary2.py $(find "$OUT/p1" -name "*counter_collection.csv"); echo "## second pass"; echo;
  python scripts/pmc_summary2.py $(find "$OUT/p2" -name "*counter_collection.csv"); } > "$OUT/summary.md"
grep -E "^###|^derived" "$OUT/summary.md"
