#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/t_b2.log
echo "#### main"; python scripts/ablate_large.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ablate_large_main.txt
for v in np1 valu; do
  echo "#### $v"; TTX_NO_NATIVE_NODE=1 TTX_LIB=$(pwd)/fbtt-embedding_amd/variants/libttx_$v.so python scripts/ablate_large.py 2>&1 | grep -v amdgpu.ids | head -3 | tee gpurun_out/ablate_large_$v.txt
done
echo "#### main (ctypes route, for the comparison with the variants)"; TTX_NO_NATIVE_NODE=1 python scripts/ablate_large.py 2>&1 | grep -v amdgpu.ids | head -3
for v in main np1; do
  L=$(pwd)/fbtt-embedding_amd/variants/libttx_$v.so; [ $v = main ] && L=$(pwd)/fbtt-embedding_amd/libttx.so
  echo "#### cfg2 $v"; TTX_NO_NATIVE_NODE=1 TTX_LIB=$L python scripts/ablate_large.py 1 512 2>&1 | grep -v amdgpu.ids | head -3
done
echo "#### cfg4"; python scripts/ablate_large.py 1 512 4,4,8 64 2>&1 | grep -v amdgpu.ids | head -3
echo "#### r128"; python scripts/ablate_large.py 1 512 4,4,4 128 2>&1 | grep -v amdgpu.ids | head -3
