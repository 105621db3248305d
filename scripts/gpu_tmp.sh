#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_tt_gpu.py -x -q -m gpu -k "padded_shapes or four_cores" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_module_gpu.py -x -q -m gpu -k "first_factor" 2>&1 | tail -4
B="--no-cpu-baseline --no-secondary --steps 100 --warmup 20"
for w in d1024r64 cfg4; do
timeout 300 python bench.py --workload $w $B 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w', j['ms_per_step'], j['value'], j.get('kernel_us'))"
done
TTX_NO_SPLIT0=1 timeout 300 python bench.py --workload d1024r64 $B 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('d1024r64 generic', j['ms_per_step'], j['value'], j.get('kernel_us'))"
