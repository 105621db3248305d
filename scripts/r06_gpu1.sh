#!/bin/bash
# round 6, first GPU pass after the refactors: the parts of the suite that did not run yet + the new tests, the cache bandwidth
# figures (atomic against sorted), the default bench line with its new secondary list
cd /root/repo
python -m pytest tests/test_cache_gpu.py tests/test_refdev_gpu.py tests/test_tt_gpu.py tests/test_primref_gpu.py tests/test_module_gpu.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/t_part.log
python scripts/bench_cache.py > gpurun_out/cache_bw.json 2> gpurun_out/cache_bw.err
( time python bench.py ) > gpurun_out/b1.json 2> gpurun_out/b1.err
