#!/bin/bash
# A/B of the kernarg-preload build (libttx.so) against the same sources with -DTTX_NO_KHEAD and no preload flag
# (fbtt-embedding_amd/variants/libttx_nokhead.so): rocprofv3 kernel durations at cfg2, eager launches and graph replay.
W=${1:-cfg2}
for rep in 1 2; do
TTX_NO_NATIVE_NODE=1 TTX_LIB=$(pwd)/fbtt-embedding_amd/variants/libttx_nokhead.so scripts/kprof.sh nokhead$rep $W > /dev/null 2>&1
TTX_NO_NATIVE_NODE=1 scripts/kprof.sh khead$rep $W > /dev/null 2>&1
done
for t in nokhead1 khead1 nokhead2 khead2; do echo "== $t"; cat gpurun_out/kprof_$t/$W.md; done
scripts/cmp_variants.sh "--workload $W --steps 200 --warmup 20 --no-secondary" nokhead
TTX_NO_NATIVE_NODE=1 python bench.py --workload $W --steps 200 --warmup 20 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('khead', d['value'], d['ms_per_step'], d['kernel_us'])"
