#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_tt_gpu.py tests/test_module_gpu.py -x -q -m gpu -k "two_cores or four_cores or graphed_step" 2>&1 | tail -12 | tee gpurun_out/t_c5.log
scripts/kprof.sh c5 t4 t4big | grep -E "^##|ttx::|eager"
