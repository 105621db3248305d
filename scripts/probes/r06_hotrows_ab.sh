#!/bin/bash
# steady-state reduce_apply (cache rows' scatter riding in its launch) at cfg3 for builds with 64 / 32 / 16 hot rows
# (GPU box only: swaps the variant in for libttx.so in the scratch copy so that the C++ node's route is the one measured)
cp fbtt-embedding_amd/libttx.so /tmp/libttx_base.so
for v in base hot32 hot16 base; do
  if [ $v = base ]; then cp /tmp/libttx_base.so fbtt-embedding_amd/libttx.so; else cp fbtt-embedding_amd/variants/libttx_$v.so fbtt-embedding_amd/libttx.so; fi
  scripts/kprof.sh hr_$v cfg3 > /dev/null 2>&1
  python - $v <<'P'
import csv,glob,sys,statistics
v=sys.argv[1]
f=glob.glob(f'gpurun_out/kprof_hr_{v}/cfg3/runc/*kernel_trace.csv')[0]
for name in ('reduce_apply','pool4_small_cached','spec_bwd','spec_fwd'):
    rows=[r for r in csv.DictReader(open(f)) if name in r['Kernel_Name']]
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows][-100:]
    if d: print(v, name, 'steady median %.2f us  min %.2f' % (statistics.median(d), min(d)), 'grid', rows[-1]['Grid_Size_X'])
P
done
cp /tmp/libttx_base.so fbtt-embedding_amd/libttx.so
