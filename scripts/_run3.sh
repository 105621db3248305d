cd /root/repo
python scripts/bench_cache.py > gpurun_out/cache_bw_new.json 2> gpurun_out/cache_bw_new.err
python -m pytest tests/test_cache_gpu.py tests/test_refdev_gpu.py tests/test_cfg3_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/t_cache.log
python bench.py --workload cfg3 --steps 100 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/b_cfg3.json
python bench.py --steps 100 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/b_cfg2.json
