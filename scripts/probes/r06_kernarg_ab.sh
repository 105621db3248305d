mkdir -p gpurun_out/exp1
for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > gpurun_out/exp1/kernarg$v.json 2> gpurun_out/exp1/kernarg$v.err
  python - <<P
import json
d=json.loads(open('gpurun_out/exp1/kernarg$v.json').read().strip().splitlines()[-1])
print('KERNARG=$v', d['ms_per_step'], d['no_prefetch']['ms_per_step'], d['eager_ms_per_step'], d['kernel_us'])
P
done
