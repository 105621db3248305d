#!/bin/bash
# the wide-digit plan route with / without the pivot's hot-slice count (skewed large batches: who sums a hot pivot slice)
cp fbtt-embedding_amd/libttx.so /tmp/libttx_base.so
for v in base pvunk; do
  if [ $v = base ]; then cp /tmp/libttx_base.so fbtt-embedding_amd/libttx.so; else cp fbtt-embedding_amd/variants/libttx_$v.so fbtt-embedding_amd/libttx.so; fi
  for w in tb4z cfg5shard t2; do
  scripts/kprof.sh wh_$v $w > /dev/null 2>&1
  grep "reduce_apply" gpurun_out/kprof_wh_$v/$w.md | sed "s/^/$v $w /"
  done
done
cp /tmp/libttx_base.so fbtt-embedding_amd/libttx.so
