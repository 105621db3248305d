#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/t_b7.log
python scripts/phase_times_large.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/phase_large_b7.txt
scripts/kprof.sh b7 cfg5shard tb4 cfg2 cfg5full | grep -E "^##|ttx::|eager"
