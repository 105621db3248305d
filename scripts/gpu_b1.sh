#!/bin/bash
mkdir -p gpurun_out
build/mfma4x4_probe > gpurun_out/mfma4x4.txt 2>&1; tail -3 gpurun_out/mfma4x4.txt
python scripts/ablate_large.py 2>&1 | tee gpurun_out/ablate_large_a.txt
python scripts/ablate_large.py 1 512 4,4,8 64 2>&1 | tee gpurun_out/ablate_cfg4_a.txt
