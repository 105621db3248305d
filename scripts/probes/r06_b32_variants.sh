#!/bin/bash
# round 6: bench cfg5shard on variant builds of the 32x32x2 backward (fbtt-embedding_amd/variants/libttx_<name>.so, built with -DTTX_TEST_HOOKS from build/obj_hooks).  usage: r06_b32_variants.sh name...
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for v in "$@"; do
  TTX_BWD32=${MC:-256} TTX_ALLOW_DEBUG=1 TTX_NO_NATIVE_NODE=1 TTX_LIB_HOOKS=$(pwd)/fbtt-embedding_amd/variants/libttx_$v.so timeout 600 python bench.py --workload cfg5shard --steps 40 --warmup 20 --no-cpu-baseline 2> gpurun_out/var_$v.err | tail -1 > gpurun_out/var_$v.json
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open(f'gpurun_out/var_{v}.json').read().strip().splitlines()[-1])
    print(v, d['ms_per_step'], d['kernel_us'], d['roofline']['avg_us'])
except Exception as e:
    print(v, 'ERR', e)
PY
done
