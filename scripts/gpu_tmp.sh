#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_tt_gpu.py tests/test_module_gpu.py -x -q -m gpu -k "padded_shapes or first_factor" 2>&1 | tail -12
scripts/kprof.sh d2 d768 d1024 | grep -E "^##|ttx::|eager"
echo "### generic"; KPROF_ARGS="" TTX_NO_SPLIT0=1 scripts/kprof.sh d2g d768 | grep -E "^##|ttx::|eager"
