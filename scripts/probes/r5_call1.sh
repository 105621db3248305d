#!/bin/bash
# round 5, GPU call 1: baseline of this box + reduce_apply variants (rocprofv3 per-kernel durations)
set -u
mkdir -p gpurun_out/r5c1
cp fbtt-embedding_amd/libttx.so fbtt-embedding_amd/variants/libttx_base.so
for V in base ra_pair ra_min4 ra_min2 ra_pairmin4 ra_pairmin2; do
  scripts/variants.sh r5c1 "cfg2 cfg4" $V
done > gpurun_out/r5c1/variants.txt 2>&1
for V in base ra_pair ra_pairmin2; do
  scripts/variants.sh r5c1 "cfg5shard" $V
done >> gpurun_out/r5c1/variants.txt 2>&1
grep -E "####|^## |reduce_apply|spec_bwd" gpurun_out/r5c1/variants.txt
