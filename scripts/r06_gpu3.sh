#!/bin/bash
cd /root/repo
python -m pytest tests/test_cache_gpu.py tests/test_cfg3_gpu.py -q -m gpu 2>&1 | tail -30 > gpurun_out/t_part.log
python -m pytest tests/test_module_gpu.py -q -m gpu -k "free_running or cache" 2>&1 | tail -12 >> gpurun_out/t_part.log
python scripts/bench_cache.py --only 10240,262144 > gpurun_out/cache_bw_small.json 2> gpurun_out/cache_bw_small.err
