#!/bin/bash
# Regenerate EVERY measured artefact under profiles/ from the build in this tree, in one go (GPU box, repo root):
#   scripts/regen_profiles.sh <tag>      e.g. r04
# Writes gpurun_out/profiles_<tag>/ (gpurun only brings gpurun_out/ back) -- copy its files into profiles/ afterwards.
# Order (round 4): counters and rocprof durations FIRST -- pmc_bwd_bytes.json / rocprof_kernels.json land in ./profiles on the
# box -- the bench lines LAST, so that every tracked bench line carries traffic / frac_rocprof / rocprof_avg_us of this build.
#   <tag>_kernel_stats.md       rocprofv3 --kernel-trace --stats + --pmc FETCH_SIZE / WRITE_SIZE per kernel (cfg2)
#   pmc_bwd_bytes.json          backward contraction: HBM bytes per launch + rocprof duration, stamped with the source hash
#   <tag>_sq_pmc_<w>.md         SQ + GRBM counters per kernel (MFMA busy, clock, parked / issue-stalled shares, LDS): cfg2, cfg5shard,
#                               cfg4, r128, and the generic kernels (r256, t2, t4 when present)
#   <tag>_other_workloads.md    rocprofv3 per-kernel durations of the other workloads
#   rocprof_kernels.json        the same durations as data, stamped with the source hash (bench.py reads it)
#   <tag>_cache_bandwidth.md    cache-path kernels: durations and GB/s on the algorithmic bytes
#   <tag>_bench*.json           the bench lines (default line incl. its cfg5shard `secondary` record)
set -u
TAG=${1:-r06}
REPO=$(pwd); OUT=$REPO/gpurun_out/profiles_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
PMC_W=${PMC_WORKLOADS:-"cfg2 cfg5shard cfg4 r128 r256 t4"}
KPROF_W=${KPROF_WORKLOADS:-"cfg3 cfg3warm cfg4 cfg5shard tb4 d32 d16 d256 r128 r13 d320 d512 d768 d1024 d1024r64 r256 t2 t4 t4d256 t2big"}
BENCH_W=${BENCH_WORKLOADS:-"cfg3 cfg3a105 cfg3warm cfg4 cfg5shard r128 r13 d320 d512 d768 d1024 d1024r64 t2 t4 t4d256"}
if [ -z "${REGEN_ONLY_BENCH:-}" ]; then   # (REGEN_ONLY_BENCH=1: the counters of this build are in ./profiles already -- bench lines only)
scripts/measure_traffic.sh "$TAG" > "$OUT/measure_traffic.log" 2>&1
cp gpurun_out/prof_$TAG/${TAG}_kernel_stats.md gpurun_out/prof_$TAG/pmc_bwd_bytes.json "$OUT/" 2>/dev/null
TRAFFIC_W=${TRAFFIC_WORKLOADS:-"cfg3 cfg4 cfg5shard"}
for W in $TRAFFIC_W; do   # (round 5: HBM bytes for the workloads where bytes are the argument)
  scripts/measure_traffic.sh "$TAG" --workload $W > "$OUT/measure_traffic_$W.log" 2>&1
  cp gpurun_out/prof_${TAG}_$W/${TAG}_kernel_stats_$W.md "$OUT/" 2>/dev/null
done
cp profiles/pmc_bytes.json "$OUT/" 2>/dev/null
for W in $PMC_W; do
  scripts/pmc_large.sh "$TAG" $W > "$OUT/pmc_$W.log" 2>&1
  cp gpurun_out/pmc_${TAG}_$W/summary.md "$OUT/${TAG}_sq_pmc_$W.md" 2>/dev/null
done
{ echo "# $TAG: rocprofv3 --kernel-trace --stats per-kernel durations of the other workloads (eager launches, 30 steps; MI355X)"; echo;
  scripts/kprof.sh "$TAG" $KPROF_W 2>&1; } > "$OUT/${TAG}_other_workloads.md"
cp gpurun_out/kprof_$TAG/rocprof_kernels.json "$OUT/" 2>/dev/null
cp gpurun_out/kprof_$TAG/rocprof_kernels.json profiles/ 2>/dev/null   # (on the box: the bench lines below read it)
scripts/cache_rocprof.sh "$TAG" > /dev/null 2>&1
cp gpurun_out/cache_prof_$TAG/summary.md "$OUT/${TAG}_cache_bandwidth.md" 2>/dev/null
fi
python bench.py --steps 200 --repeats 5 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
for W in $BENCH_W; do
  python bench.py --workload $W --steps 100 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_$W.json"
done
ls -la "$OUT"
