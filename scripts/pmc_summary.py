#!/usr/bin/env python3
"""Average rocprofv3 PMC counters per kernel from *_counter_collection.csv files."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("ttx::", "")
        if "ttx" not in r["Kernel_Name"]:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:34s} {sum(v)/len(v):14.1f}  (n={len(v)})")
