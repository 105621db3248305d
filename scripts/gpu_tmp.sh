#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_module_gpu.py tests/test_cfg3_gpu.py -x -q -m gpu -k "cache or cfg3 or live or populate" 2>&1 | tail -4
B="--no-cpu-baseline --no-secondary --steps 100 --warmup 10 --repeats 3"
for i in 1 2; do
timeout 300 python bench.py --workload cfg3 $B 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg3 fused', j['ms_per_step'], j['no_prefetch']['ms_per_step'], j['eager_ms_per_step'], j.get('kernel_us'))"
TTX_NO_FUSED_CACHE_GATHER=1 timeout 300 python bench.py --workload cfg3 $B 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg3 pair ', j['ms_per_step'], j['no_prefetch']['ms_per_step'], j['eager_ms_per_step'], j.get('kernel_us'))"
done
