#!/bin/bash
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/own_prof -- python /root/repo/scripts/bench_cache.py --only 10240,262144 > /root/repo/gpurun_out/own_prof.json 2> /root/repo/gpurun_out/own_prof.err
cd /root/repo
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/own_prof/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), round(float(r["MinNs"]) / 1e3, 2), round(float(r["MaxNs"]) / 1e3, 2))
PY
