#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_tt_gpu.py -x -q -m gpu -k "four_cores" 2>&1 | tail -30 | tee gpurun_out/t_c3.log
scripts/kprof.sh c3 t4 t4big | grep -E "^##|ttx::|eager"
