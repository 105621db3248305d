#!/bin/bash
# scripts/cmp_variants.sh "<bench args>" name1 name2 ... : bench each variant build (fbtt-embedding_amd/variants/libttx_<name>.so, scripts/build_variant.sh)
ARGS=$1; shift
for v in "$@"; do
  TTX_NO_NATIVE_NODE=1 TTX_LIB=$(pwd)/fbtt-embedding_amd/variants/libttx_$v.so python bench.py $ARGS --no-cpu-baseline 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['kernel_us'])"
done
