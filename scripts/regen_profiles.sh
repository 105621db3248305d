#!/bin/bash
# Regenerate EVERY measured artefact under profiles/ from the build in this tree, in one go (GPU box, repo root):
#   scripts/regen_profiles.sh <tag>      e.g. r03
# Writes gpurun_out/profiles_<tag>/ (gpurun only brings gpurun_out/ back) -- copy its files into profiles/ afterwards.
#   <tag>_bench.json            the default bench line (cfg2, 200 steps x 5 repeats) + in-line / eager figures
#   <tag>_kernel_stats.md       rocprofv3 --kernel-trace --stats + --pmc FETCH_SIZE / WRITE_SIZE per kernel (cfg2)
#   pmc_bwd_bytes.json          backward contraction: HBM bytes per launch + rocprof duration, stamped with the source hash
#   <tag>_sq_pmc.md             SQ counters (MFMA busy, waits, LDS) of the cfg2 kernels
#   <tag>_other_workloads.md    rocprofv3 per-kernel durations of the other workloads (cfg3, cfg3warm, cfg4, cfg5shard, dims, generic)
#   rocprof_kernels.json        the same durations as data, stamped with the source hash (bench.py --workload cfg3 reads it)
#   <tag>_cache_bandwidth.md    cache-path kernels: durations and GB/s on the algorithmic bytes
set -u
TAG=${1:-r03}
REPO=$(pwd); OUT=$REPO/gpurun_out/profiles_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
python bench.py --steps 200 --repeats 5 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
scripts/measure_traffic.sh "$TAG" > "$OUT/measure_traffic.log" 2>&1
cp gpurun_out/prof_$TAG/${TAG}_kernel_stats.md gpurun_out/prof_$TAG/pmc_bwd_bytes.json "$OUT/" 2>/dev/null
{ echo "# $TAG: SQ counters of the cfg2 step's kernels (rocprofv3 --pmc, two passes of 8 counters; eager launches)"; echo;
  echo '```'; scripts/pmc_sq.sh "$TAG" 2>&1 | tail -60; echo '```'; } > "$OUT/${TAG}_sq_pmc.md"
{ echo "# $TAG: rocprofv3 --kernel-trace --stats per-kernel durations of the other workloads (eager launches, 30 steps; MI355X)"; echo;
  scripts/kprof.sh "$TAG" cfg3 cfg3warm cfg4 cfg5shard tb4 d32 d16 d256 r128 r13 d512 r256 2>&1; } > "$OUT/${TAG}_other_workloads.md"
cp gpurun_out/kprof_$TAG/rocprof_kernels.json "$OUT/" 2>/dev/null
for W in cfg3 cfg3a105 cfg3warm cfg4 cfg5shard r128 r13 d512; do
  python bench.py --workload $W --steps 100 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_$W.json"
done
scripts/cache_rocprof.sh "$TAG" > /dev/null 2>&1
cp gpurun_out/cache_prof_$TAG/summary.md "$OUT/${TAG}_cache_bandwidth.md" 2>/dev/null
ls -la "$OUT"
