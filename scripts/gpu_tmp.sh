#!/bin/bash
# scratch: the round's current GPU check
mkdir -p gpurun_out
B="--no-cpu-baseline --no-secondary --steps 200 --warmup 30"
for i in 1 2 3; do
TTX_LIB=$PWD/fbtt-embedding_amd/variants/libttx_lateb.so TTX_NO_NATIVE_NODE=1 timeout 300 python bench.py --workload cfg4 $B 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('lateb    ', j['ms_per_step'], j.get('kernel_us'))"
TTX_NO_NATIVE_NODE=1 timeout 300 python bench.py --workload cfg4 $B 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('lateb+pf ', j['ms_per_step'], j.get('kernel_us'))"
done
timeout 300 python bench.py --workload cfg4 $B 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('native   ', j['ms_per_step'], j.get('kernel_us'))"
timeout 1200 python -m pytest tests/test_tt_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -3
