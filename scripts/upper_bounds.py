#!/usr/bin/env python3
"""Upper bounds for restructurings of the cfg2 step (timing only -- results are INVALID under these masks):
what the captured step costs without the pooling launch, without reduce_apply, without the pivot partials."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASKS = [("baseline", 0), ("no pool launch", 512), ("no reduce_apply launch", 1024), ("no pivot partials (store + reduce)", 2048 + 128),
         ("no pool, no pivot partials", 512 + 2048 + 128), ("no pool, no reduce_apply", 512 + 1024)]
for name, mask in MASKS:
    env = dict(os.environ, TTX_DEBUG_SKIP=str(mask), TTX_ALLOW_DEBUG="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "200", "--repeats", "3"],
                         capture_output=True, text=True, env=env)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"{name:40s} planned {j['ms_per_step']:.4f}  in-line {j['no_prefetch']['ms_per_step'] if j.get('no_prefetch') else None}  eager {j['eager_ms_per_step']:.4f}", flush=True)
    except Exception as ex:  # noqa: BLE001
        print(name, "FAILED", ex, out.stderr[-500:], flush=True)
