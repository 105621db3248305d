#!/bin/bash
# round 5, GPU call 2: slice owners of d core_1 -- parity, then A/B (TTX_DEBUG_SKIP=65536 = owners off)
set -u
mkdir -p gpurun_out/r5c2
timeout 900 python -m pytest tests/test_tt_gpu.py tests/test_module_gpu.py -x -q -m gpu > gpurun_out/r5c2/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r5c2/pytest.log
{ echo "#### owners ON"; scripts/kprof.sh r5c2on cfg2 cfg4 cfg5shard d256 2>&1 | grep -E "^## |spec_bwd|reduce_apply|eager";
  echo "#### owners OFF"; TTX_DEBUG_SKIP=65536 scripts/kprof.sh r5c2off cfg2 cfg4 cfg5shard d256 2>&1 | grep -E "^## |spec_bwd|reduce_apply|eager"; } | tee gpurun_out/r5c2/ab.txt
for m in 0 65536; do TTX_DEBUG_SKIP=$m python bench.py --steps 200 --repeats 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('mask $m', j['ms_per_step'], j['no_prefetch'], j['eager_ms_per_step'])"; done | tee gpurun_out/r5c2/bench_ab.txt
