#!/bin/bash
# rocprofv3 passes over the default bench workload (run on the GPU box from the repo root):
#   1. --kernel-trace --stats          -> per-kernel durations
#   2. --pmc FETCH_SIZE                -> HBM read bytes   (own pass, see MI355X_MICROARCH.md "HBM")
#   3. --pmc WRITE_SIZE                -> HBM write bytes  (own pass)
# usage: scripts/profile_round.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 50 --warmup 10 --no-cpu-baseline $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python "$REPO/bench.py" $ARGS > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_rd" -- python "$REPO/bench.py" $ARGS --no-graph > "$OUT/pmc_rd.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_wr" -- python "$REPO/bench.py" $ARGS --no-graph > "$OUT/pmc_wr.log" 2>&1
cd "$REPO"
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -r head -25
python scripts/pmc_summary.py $(find "$OUT" -name "*counter_collection.csv") | tee "$OUT/pmc_summary.txt"
tail -1 "$OUT/trace.log"
