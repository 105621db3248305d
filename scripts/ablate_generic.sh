#!/bin/bash
# where the generic (block-walk) kernels' time goes: the r128 / r13 workloads with kernel phases left out (timing only)
for w in ${1:-r128}; do
for m in 0 1 2 4 8 15; do
TTX_ALLOW_DEBUG=1 TTX_DEBUG_SKIP=$m python bench.py --workload $w --steps 20 --repeats 2 --no-cpu-baseline --no-graph 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$w mask $m (1 = no x0 GEMM, 2 = no tail, 4 = no d core_1, 8 = no d core_0)', d['kernel_us'])"
done; done
