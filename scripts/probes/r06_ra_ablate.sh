#!/bin/bash
# where reduce_apply's time goes on two cores (t2: 2 x 3317 slices of 256 floats, wave-per-slice path): the launch as built, cut
# behind the slice offsets (abl1), cut in front of the apply (abl2) -- timing only, the cut builds' results are invalid
cp fbtt-embedding_amd/libttx.so /tmp/libttx_base.so
for v in base abl1 abl2; do
  if [ $v = base ]; then cp /tmp/libttx_base.so fbtt-embedding_amd/libttx.so; else cp fbtt-embedding_amd/variants/libttx_$v.so fbtt-embedding_amd/libttx.so; fi
  for w in t2 cfg2; do
  scripts/kprof.sh ra_$v $w > /dev/null 2>&1
  grep "reduce_apply" gpurun_out/kprof_ra_$v/$w.md | sed "s/^/$v $w /"
  done
done
cp /tmp/libttx_base.so fbtt-embedding_amd/libttx.so
