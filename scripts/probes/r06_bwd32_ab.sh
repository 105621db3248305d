#!/bin/bash
# round 6: A/B of the 32x32x2 backward (test build, ttx_debug_bwd32 = TTX_BWD32=<lookups per chunk> through the shim) on the large-batch
# workload + its parity tests.  gpurun -- bash scripts/probes/r06_bwd32_ab.sh
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
TTX_BWD32=128 timeout 900 python -m pytest tests/test_tt_gpu.py tests/test_cfg5_gpu.py -k "large_batch or cfg5 or bwd32" -x -q > gpurun_out/t_b32.log 2>&1
tail -3 gpurun_out/t_b32.log
for mc in ${MCS:-128}; do
  TTX_BWD32=$mc TTX_ALLOW_DEBUG=1 timeout 600 python bench.py --workload cfg5shard --steps 40 --warmup 20 --no-cpu-baseline > gpurun_out/b32_mc$mc.json 2> gpurun_out/b32_mc$mc.err
done
timeout 600 python bench.py --workload cfg5shard --steps 40 --warmup 20 --no-cpu-baseline > gpurun_out/b32_off.json 2> gpurun_out/b32_off.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/b32_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['kernel_us'], d['roofline']['avg_us'])
    except Exception as e:
        print(f, 'ERR', e)
PY
