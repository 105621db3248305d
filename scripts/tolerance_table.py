#!/usr/bin/env python3
"""TTX_TOL_REPORT records (tests/util.py) -> the markdown table of DESIGN.md section 5: every comparison wider than the default
tolerance (rtol 1e-5, atol 2e-6 max|ref|), its bound, and the worst error the GPU suite actually measured under it.
usage: scripts/tolerance_table.py gpurun_out/tolerances.jsonl > profiles/r05_tolerances.md"""
import collections
import json
import sys

rows = collections.OrderedDict()
for ln in open(sys.argv[1]):
    j = json.loads(ln)
    k = (j["test"], j["what"])
    if k not in rows or j["worst_over_default"] > rows[k]["worst_over_default"]:
        c = rows[k]["calls"] if k in rows else 0
        rows[k] = j
        rows[k]["calls"] = c + j["calls"]
print("| test | comparison | rtol | atol / max\\|ref\\| | worst error measured, in units of the DEFAULT bound (1e-5, 2e-6) | share of the widened bound used | calls |")
print("|---|---|---|---|---|---|---|")
for (t, w), j in sorted(rows.items()):
    print(f"| `{t.replace('tests/', '')}` | {w} | {j['rtol']:g} | {j['atol_scale']:g} | {j['worst_over_default']:.2f} | {j['share_of_bound_used']:.3f} | {j['calls']} |")
