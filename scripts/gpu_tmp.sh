#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_tt_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -5
