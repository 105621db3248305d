#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_tt_gpu.py -x -q -m gpu -k "four_cores" 2>&1 | tail -5
B="--no-cpu-baseline --no-secondary --steps 100 --warmup 20"
timeout 300 python bench.py --workload t4d256 $B 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('t4d256', j['ms_per_step'], j['value'], j.get('kernel_us'))"
TTX_DEBUG_SKIP=256 timeout 300 python bench.py --workload t4d256 $B 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('t4d256 generic?', j['ms_per_step'], j['value'], j.get('kernel_us'))"
