#!/bin/bash
# rocprofv3 kernel durations of the cache-path kernels (gather fwd, SGD scatter bwd, hash update, lookup + partition) at
# one size per run -> gpurun_out/cache_prof_<tag>/summary.md with the achieved GB/s on the ALGORITHMIC bytes of
# SURVEY.md section 8(d).  usage (GPU box, repo root): scripts/cache_rocprof.sh <tag>
set -u
TAG=${1:-r02}; REPO=$(pwd); OUT=$REPO/gpurun_out/cache_prof_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for CFG in 10240,262144 1048576,262144 1048576,4194304; do
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$CFG" -- python "$REPO/scripts/bench_cache.py" --only $CFG > "$OUT/$CFG.json" 2> "$OUT/$CFG.err"
  cd "$REPO"
done
python - "$OUT" <<'PY' | tee "$OUT/summary.md"
import csv, glob, json, os, sys
out = sys.argv[1]
D, L = 64, 20
print("# cache path: rocprofv3 --kernel-trace --stats durations per launch and achieved bandwidth on the algorithmic bytes (MI355X)\n")
print("Algorithmic bytes (SURVEY.md 8d): gather fwd 4D+12 = 268 B per cached lookup + 4D per bag; SGD scatter 2*4D+12 = 524 B per cached "
      "lookup + 4D per bag; hash update 24 B per key.  D = 64, 20 lookups per bag.  Peak 8 TB/s HBM3E (6.3 TB/s achievable streaming).\n")
print("| lookups | cache rows (MiB) | kernel | calls | avg us | GB/s | of 8 TB/s |")
print("|---|---|---|---|---|---|---|")
for cfg in ("10240,262144", "1048576,262144", "1048576,4194304"):
    nnz, rows = (int(x) for x in cfg.split(","))
    B = nnz // L
    nnz = B * L
    byt = {"cache_forward4_kernel": nnz * (4 * D + 12) + B * 4 * D, "cache_forward_kernel": nnz * (4 * D + 12) + B * 4 * D,
           "cache_scatter_add_kernel": nnz * (2 * 4 * D + 12) + B * 4 * D, "update_cache_state_kernel": nnz * 24}
    for f in glob.glob(os.path.join(out, cfg, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Name"].split("(")[0].replace("void ", "").replace("ttx::", "")
            if name in byt:
                us = float(r["AverageNs"]) / 1e3
                gbs = byt[name] / us / 1e3
                print(f"| {nnz} | {rows} ({rows * D * 4 >> 20}) | `{name}` | {r['Calls']} | {us:.2f} | {gbs:.0f} | {gbs / 80:.1f} % |")
# (round 6) the atomic-free update is a chain of launches: every kernel of the run, so that the chain can be added up by hand
# (bench_cache.py times each variant 23 times: calls / 23 = launches of that kernel per update, summed over the variants that use it)
print("\n## every kernel of the runs (the sorted update = cs_keys + radix_* x passes + dd_* + gsum_* (+ cs_scan_* / cs_bag_g2 / cs_state for Adagrad) + cs_apply)\n")
for cfg in ("10240,262144", "1048576,262144", "1048576,4194304"):
    print(f"\n### {cfg}\n\n| kernel | calls | avg us | total us |\n|---|---|---|---|")
    rowsx = []
    for f in glob.glob(os.path.join(out, cfg, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Name"].split("(")[0].replace("void ", "").replace("ttx::", "")
            rowsx.append((float(r["TotalDurationNs"]) / 1e3, name, int(r["Calls"]), float(r["AverageNs"]) / 1e3))
    for tot, name, calls, avg in sorted(rowsx, reverse=True)[:24]:
        print(f"| `{name[:90]}` | {calls} | {avg:.2f} | {tot:.0f} |")
PY
