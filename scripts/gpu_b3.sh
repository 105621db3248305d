#!/bin/bash
mkdir -p gpurun_out
python scripts/phase_times_large.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/phase_large_main.txt
echo "#### main (ctypes)"; TTX_NO_NATIVE_NODE=1 python scripts/ablate_large.py 2>&1 | grep -v amdgpu.ids | head -2
for v in "$@"; do
  echo "#### $v"; TTX_NO_NATIVE_NODE=1 TTX_LIB=$(pwd)/fbtt-embedding_amd/variants/libttx_$v.so python scripts/ablate_large.py 2>&1 | grep -v amdgpu.ids | head -2 | tee gpurun_out/ablate_large_$v.txt
done
