#!/usr/bin/env python3
"""rocprofv3 outputs of scripts/measure_traffic.sh -> profiles/<tag>_kernel_stats.md (durations + HBM bytes per
launch of every ttx kernel) and, for the default workload, profiles/pmc_bwd_bytes.json (the backward
contraction's HBM bytes per launch, stamped with the hash of the kernel sources).
usage: pmc_to_json.py <tag> <gpurun_out/prof_tag> [bench args]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, out = sys.argv[1], sys.argv[2]
bench_args = sys.argv[3] if len(sys.argv) > 3 else ""


def short(name):
    return name.split("(")[0].replace("void ", "").replace("ttx::", "")


counters = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "ttx" in r["Kernel_Name"]:
            counters[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
stats = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "ttx::" in r["Name"]:
            stats[short(r["Name"])] = r
avg = lambda v: sum(v) / len(v) if v else None  # noqa: E731
lines = [f"# {tag}: rocprofv3 per-kernel durations (--kernel-trace --stats) and HBM bytes per launch (--pmc FETCH_SIZE / --pmc "
         f"WRITE_SIZE, separate passes) of `python bench.py {bench_args}` on MI355X\n",
         "FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE as reported.\n",
         "| kernel | calls | avg us | min us | max us | % of GPU time | HBM read KiB | HBM written KiB | GB/s (read+write) |",
         "|---|---|---|---|---|---|---|---|---|"]
tot_us, tot_b = 0.0, 0.0
for k, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["Percentage"])):
    rd, wr = avg(counters[k].get("FETCH_SIZE", [])), avg(counters[k].get("WRITE_SIZE", []))
    us = float(r["AverageNs"]) / 1e3
    b = ((rd or 0) * 2 + (wr or 0)) * 1024
    tot_us += us
    tot_b += b
    lines.append(f"| `{k}` | {r['Calls']} | {us:.2f} | {int(r['MinNs']) / 1e3:.2f} | {int(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.1f} | "
                 f"{'' if rd is None else f'{2 * rd:.0f}'} | {'' if wr is None else f'{wr:.0f}'} | {b / us / 1e3 if rd is not None else 0:.0f} |")
lines.append(f"\nsum of the kernel averages: {tot_us:.1f} us per step; HBM bytes per step (read x2 + written): {tot_b / 1e6:.1f} MB")
log = os.path.join(out, "trace.log")
if os.path.exists(log):
    last = open(log).read().strip().splitlines()[-1]
    if last.startswith("{"):
        j = json.loads(last)
        lines.append(f"\nbench line of the same run (under rocprofv3): {j['value']} GFLOP/s, {j['ms_per_step']} ms/step ({j['timed_mode']}), "
                     f"eager {j['eager_ms_per_step']} ms/step; workload: {j['config']['workload']}")
for d in (os.path.join(ROOT, "profiles"), out):  # (gpurun only brings gpurun_out/ back: copy from there into profiles/)
    open(os.path.join(d, f"{tag}_kernel_stats{'' if '--workload' not in bench_args else '_' + bench_args.split()[bench_args.split().index('--workload') + 1]}.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

# every workload: profiles/pmc_bytes.json -- HBM bytes per launch and the rocprofv3 duration of every ttx kernel, per workload,
# stamped with the hash of the kernel sources (bench.py reports roofline.traffic / rocprof_avg_us from it for the build it matches)
import bench  # noqa: E402

wl = "cfg2"
toks = bench_args.split()
if "--workload" in toks:
    wl = toks[toks.index("--workload") + 1]
blob_all = {"source_hash": bench.source_hash(), "fetch_correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B, "
            "MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; KiB -> bytes x1024; separate --pmc passes", "workloads": {}}
for d in (os.path.join(ROOT, "profiles"), out):
    fn = os.path.join(d, "pmc_bytes.json")
    cur = blob_all
    if os.path.exists(fn):
        try:
            j = json.load(open(fn))
            if j.get("source_hash") == blob_all["source_hash"]:
                cur = j
        except Exception:  # noqa: BLE001
            pass
    ks = {}
    for k in counters:
        rd, wr = avg(counters[k].get("FETCH_SIZE", [])), avg(counters[k].get("WRITE_SIZE", []))
        if rd is None or wr is None:
            continue
        ks[k] = {"read_bytes": int(2 * rd * 1024), "write_bytes": int(wr * 1024), "hbm_bytes_per_launch": int((2 * rd + wr) * 1024),
                 "pmc_launches": len(counters[k]["FETCH_SIZE"]),
                 "rocprof_avg_us": round(float(stats[k]["AverageNs"]) / 1e3, 3) if k in stats else None,
                 "rocprof_calls": int(stats[k]["Calls"]) if k in stats else 0}
    cur["workloads"][wl] = {"kernels": ks, "source": f"scripts/measure_traffic.sh {tag} {bench_args}: rocprofv3 --kernel-trace --stats "
                            f"(durations), --pmc FETCH_SIZE and --pmc WRITE_SIZE in passes of their own (--no-graph)"}
    json.dump(cur, open(fn, "w"), indent=1)
print("wrote profiles/pmc_bytes.json for", wl)

bwd = [k for k in counters if "bwd_kernel" in k]
if bwd and "--workload" not in bench_args:
    k = bwd[0]
    rd, wr = avg(counters[k].get("FETCH_SIZE", [])), avg(counters[k].get("WRITE_SIZE", []))
    if rd is not None and wr is not None:
        blob = {
            "kernel": k,
            "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py {bench_args} --no-graph "
                      f"(scripts/measure_traffic.sh {tag}), per-launch averages over {len(counters[k]['FETCH_SIZE'])} launches",
            "source_hash": bench.source_hash(),
            "fetch_size_kib_raw": round(rd, 1), "write_size_kib": round(wr, 1),
            "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section)",
            "hbm_bytes_per_launch": int((2 * rd + wr) * 1024),
            "rocprof_avg_us": round(float(stats[k]["AverageNs"]) / 1e3, 3) if k in stats else None,
            "rocprof_source": f"rocprofv3 --kernel-trace --stats of bench.py {bench_args} (hipGraph replay), "
                              f"{stats[k]['Calls'] if k in stats else 0} launches",
        }
        for d in (os.path.join(ROOT, "profiles"), out):
            json.dump(blob, open(os.path.join(d, "pmc_bwd_bytes.json"), "w"), indent=1)
        print("wrote profiles/pmc_bwd_bytes.json")
