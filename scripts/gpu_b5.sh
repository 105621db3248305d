#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/t_b5.log
python scripts/ablate_large.py 2>&1 | grep -v amdgpu.ids | head -3 | tee gpurun_out/ablate_large_b5.txt
python scripts/phase_times_large.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/phase_large_b5.txt
scripts/kprof.sh b5 cfg5shard cfg2 | grep -E "^##|ttx::|eager"
