#!/bin/bash
mkdir -p gpurun_out
B="--no-cpu-baseline --no-secondary --steps 100 --warmup 10 --repeats 1"
for w in x_q12 x_q16r24 x_q0_3 r13 t4 t4d256 d768 cfg2; do
timeout 300 python bench.py --workload $w $B 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w', j['ms_per_step'], j.get('kernel_us'))"
done
timeout 1200 python -m pytest tests/test_tt_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -3
