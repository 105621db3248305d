#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 | tee gpurun_out/t_full.log
