#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 PMC counters (*_counter_collection.csv) WITH the dispatch duration and the figures
derived from them (MI355X_MICROARCH.md, "rocprofv3 PMC slots"):
  clock      = GRBM_GUI_ACTIVE / 8 XCDs / duration             (effective shader clock while the kernel ran; the counter sums the XCDs)
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)  (share of the run the matrix pipes were busy)
  parked / issue-stall / active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
usage: pmc_summary2.py <csv> [<csv> ...]      (markdown on stdout, ttx kernels only)"""
import collections
import csv
import sys

CLOCK_LO, CLOCK_HI, CLOCK_LONG = 2.0, 2.6, 2.4  # GHz: plausible shader clocks; what long kernels measure (r04: 2.35 .. 2.43)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
seen = set()
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        if "ttx::" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ttx::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in acc.items():
    d = sum(dur[k]) / len(dur[k])
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print(f"### `{k}`  ({len(dur[k])} dispatches, avg {d:.1f} us under the counters)")
    print("| counter | avg per dispatch |\n|---|---|")
    for c in sorted(m):
        print(f"| {c} | {m[c]:.0f} |")
    der = []
    gui = m.get("GRBM_GUI_ACTIVE")
    if gui:
        clock = gui / 8 / d / 1e3
        if CLOCK_LO <= clock <= CLOCK_HI:
            run_cycles = gui / 8  # per XCD
            der.append(f"clock {clock:.2f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)")
            how = "busy cycles / (1024 SIMDs x GUI_ACTIVE / 8 XCDs)"
        else:
            # GRBM_GUI_ACTIVE includes the counters' start / stop time around a short dispatch: the "clock" it implies is not one
            # (round 4 printed 3.2 .. 7.1 GHz for kernels under ~30 us and understated everything divided by it).  Long kernels
            # of the same runs come out at 2.35 .. 2.43 GHz: the run's cycles are taken as duration x that clock instead.
            run_cycles = d * 1e3 * CLOCK_LONG
            der.append(f"(GRBM_GUI_ACTIVE / 8 / duration = {clock:.2f} GHz is outside {CLOCK_LO}-{CLOCK_HI} GHz: the counter includes its "
                       f"own start / stop time around a {d:.1f} us dispatch -- run cycles taken as duration x {CLOCK_LONG} GHz)")
            how = f"busy cycles / (1024 SIMDs x duration x {CLOCK_LONG} GHz)"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * run_cycles)
            der.append(f"MFMA pipes busy {busy:.3f} of the run ({how})")
    if "SQ_WAVE_CYCLES" in m:
        w = m["SQ_WAVE_CYCLES"]
        for c, nm in (("SQ_WAIT_ANY", "parked (waitcnt / barrier)"), ("SQ_WAIT_INST_ANY", "issue-stalled"), ("SQ_ACTIVE_INST_ANY", "issuing")):
            if c in m:
                der.append(f"{nm} {m[c] / w:.3f} of wave time")
    if "SQ_INSTS_VALU_MFMA_MOPS_F32" in m and "SQ_INSTS_VALU" in m:
        der.append(f"VALU instructions per 512 MFMA MOPS {m['SQ_INSTS_VALU'] / max(m['SQ_INSTS_VALU_MFMA_MOPS_F32'] / 512, 1):.2f}")
    if der:
        print("\nderived: " + "; ".join(der))
    print()
