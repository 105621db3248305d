#!/bin/bash
mkdir -p gpurun_out
for L in 0 70000 100000; do
  echo "######## TTX_DEBUG_BWD_LDS=$L"
  TTX_DEBUG_BWD_LDS=$L python scripts/ablate_large.py 2>&1 | grep -v amdgpu.ids | head -2
done
