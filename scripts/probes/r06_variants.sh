#!/bin/bash
# round 6: bench a workload on variant builds of the PRODUCT library (fbtt-embedding_amd/variants/libttx_<name>.so, scripts/build_variant_fast.sh).
# usage: W=cfg5shard scripts/probes/r06_variants.sh name...
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for v in "$@"; do
  TTX_NO_NATIVE_NODE=1 TTX_LIB=$(pwd)/fbtt-embedding_amd/variants/libttx_$v.so timeout 600 python bench.py --workload ${W:-cfg5shard} --no-secondary --no-cpu-baseline 2> gpurun_out/pv_$v.err | tail -1 > gpurun_out/pv_$v.json
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open(f'gpurun_out/pv_{v}.json').read().strip().splitlines()[-1])
    print(v, d['ms_per_step'], d['kernel_us'], d['roofline']['avg_us'])
except Exception as e:
    print(v, 'ERR', e)
PY
done
