#!/bin/bash
# scripts/cmp_variants.sh name1 name2 ... : bench each variant (hipGraph value + kernel breakdown)
for v in "$@"; do
  TTX_LIB=$(pwd)/variants/libttx_$v.so python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['kernel_us'])"
done
