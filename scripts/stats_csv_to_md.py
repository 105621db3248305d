#!/usr/bin/env python3
"""rocprofv3 --stats kernel_stats.csv (+ the PMC summary) -> a markdown table for profiles/.
usage: stats_csv_to_md.py <kernel_stats.csv> [title] > profiles/xxx.md"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
title = sys.argv[2] if len(sys.argv) > 2 else "rocprofv3 --kernel-trace --stats"
print(f"# {title}\n")
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
for r in rows:
    n = r["Name"]
    n = n.split("(")[0].replace("void ", "") if ("ttx::" in n or len(n) > 90) else n
    n = n[:90]
    print(f"| `{n}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | "
          f"{int(r['MinNs']) / 1e3:.2f} | {int(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.1f} |")
