#!/bin/bash
# scratch: the round's current GPU check
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_module_gpu.py -x -q -m gpu -k "padded_first" 2>&1 | tail -15
