#!/bin/bash
cp fbtt-embedding_amd/libttx.so /tmp/libttx_base.so
for v in base abl1 abl3 abl4; do
  if [ $v = base ]; then cp /tmp/libttx_base.so fbtt-embedding_amd/libttx.so; else cp fbtt-embedding_amd/variants/libttx_$v.so fbtt-embedding_amd/libttx.so; fi
  for w in cfg2 cfg5shard; do
  scripts/kprof.sh ra2_$v $w > /dev/null 2>&1
  grep "reduce_apply" gpurun_out/kprof_ra2_$v/$w.md | sed "s/^/$v $w /"
  done
done
cp /tmp/libttx_base.so fbtt-embedding_amd/libttx.so
