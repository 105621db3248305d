#!/bin/bash
mkdir -p gpurun_out
B="--no-cpu-baseline --no-secondary --steps 100 --warmup 10 --repeats 1"
for w in t4 t4d256; do
for sg in 32 16 8 4; do
TTX_T4_SEG=$sg timeout 300 python bench.py --workload $w $B 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w seg=$sg', j['ms_per_step'], j.get('kernel_us'))"
done
done
timeout 900 python -m pytest tests/test_tt_gpu.py -x -q -m gpu -k "four_cores" 2>&1 | tail -2
