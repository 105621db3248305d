#!/bin/bash
# HBM traffic per launch of every kernel of the bench workload, and the JSON bench.py's roofline.traffic reads.
# Run on the GPU box from the repo root:   scripts/measure_traffic.sh <tag> [bench args...]
#   1. rocprofv3 --kernel-trace --stats   -> per-kernel durations            -> gpurun_out/prof_<tag>/trace
#   2. rocprofv3 --pmc FETCH_SIZE         -> HBM read bytes  (own pass: MI355X_MICROARCH.md "HBM" / "PMC slots")
#   3. rocprofv3 --pmc WRITE_SIZE         -> HBM write bytes (own pass)
# then scripts/pmc_to_json.py rewrites profiles/pmc_bwd_bytes.json (stamped with the kernel sources' hash: bench.py
# reports the figure only for the build it was measured on) and writes profiles/<tag>_kernel_stats.md.
set -u
TAG=${1:-r02}; shift || true
REPO=$(pwd)
WSUF=""   # (--workload W: a directory of its own, profiles/<tag>_kernel_stats_W.md; every workload lands in profiles/pmc_bytes.json)
prev=""; for a in "$@"; do [ "$prev" = "--workload" ] && WSUF="_$a"; prev=$a; done
OUT=$REPO/gpurun_out/prof_$TAG$WSUF
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 50 --warmup 10 --repeats 1 --no-cpu-baseline --no-secondary $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python "$REPO/bench.py" $ARGS > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_rd" -- python "$REPO/bench.py" $ARGS --no-graph > "$OUT/pmc_rd.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_wr" -- python "$REPO/bench.py" $ARGS --no-graph > "$OUT/pmc_wr.log" 2>&1
cd "$REPO"
python scripts/pmc_to_json.py "$TAG" "$OUT" "$ARGS"
tail -1 "$OUT/trace.log" | cut -c1-600
