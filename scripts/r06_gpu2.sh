#!/bin/bash
cd /root/repo
python -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/t_full.log
scripts/cache_rocprof.sh r06 > /dev/null 2>&1
