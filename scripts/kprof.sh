#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of one or more bench workloads, ttx kernels only.
# usage (GPU box, repo root): scripts/kprof.sh <tag> <workload> [<workload> ...]   -> gpurun_out/kprof_<tag>/<workload>.md
set -u
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/kprof_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for W in "$@"; do
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$W" -- python "$REPO/bench.py" --workload "$W" --steps 30 --warmup 5 \
      --repeats 1 --no-cpu-baseline --no-secondary --no-graph ${KPROF_ARGS:-} > "$OUT/$W.log" 2>&1
  cd "$REPO"
  F=$(find "$OUT/$W" -name "*kernel_stats.csv" | head -1)
  { echo "## $W"; python scripts/stats_csv_to_md.py "$F" "$W" | grep -E "ttx::|^\| kernel|^\|---"; grep "^{\"metric" "$OUT/$W.log" | tail -1 | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read()); print('eager ms/step', j['eager_ms_per_step'], '| kernel_us (event brackets)', j['kernel_us'])
except Exception as e: print('no bench line', e)
"; } | tee "$OUT/$W.md"
  # the same averages as data (bench.py reads profiles/rocprof_kernels.json for its rocprof-derived fractions)
  python - "$F" "$W" "$OUT/rocprof_kernels.json" <<'PY'
import csv, json, os, sys
sys.path.insert(0, os.getcwd())
import bench
f, w, out = sys.argv[1:4]
fresh = {"source_hash": bench.source_hash(), "how": "rocprofv3 --kernel-trace --stats, bench.py --workload W --steps 30 --no-graph (scripts/kprof.sh)", "workloads": {}}
j = json.load(open(out)) if os.path.exists(out) else fresh
if j.get("source_hash") != fresh["source_hash"]:
    j = fresh
ks = {}
for r in csv.DictReader(open(f)):
    if "ttx::" in r["Name"]:
        ks[r["Name"].split("(")[0].replace("void ", "").replace("ttx::", "")] = {"avg_us": round(float(r["AverageNs"]) / 1e3, 3), "calls": int(r["Calls"])}
j["workloads"][w] = ks
json.dump(j, open(out, "w"), indent=1)
PY
done
