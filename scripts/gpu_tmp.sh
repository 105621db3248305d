#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/t_full.log
scripts/regen_profiles.sh r04 2>&1 | tail -3
