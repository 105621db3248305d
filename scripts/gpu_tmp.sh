#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_tt_gpu.py -x -q -m gpu -k "two_cores or four_cores" 2>&1 | tail -30 | tee gpurun_out/t_c4.log
scripts/kprof.sh c4 t2 t2big | grep -E "^##|ttx::|eager"
