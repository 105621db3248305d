#!/bin/bash
mkdir -p gpurun_out
python scripts/phase_times_large.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/phase_large_b6.txt
echo "######## one work-group per CU"
TTX_DEBUG_BWD_LDS=100000 python scripts/phase_times_large.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/phase_large_b6_1wg.txt
echo "######## two work-groups per CU"
TTX_DEBUG_BWD_LDS=70000 python scripts/phase_times_large.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/phase_large_b6_2wg.txt
python scripts/ablate_large.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ablate_large_b6.txt
