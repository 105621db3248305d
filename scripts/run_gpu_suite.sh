#!/bin/bash
# the whole -m gpu suite + smoke() on the GPU box: gpurun -- bash scripts/run_gpu_suite.sh  ->  gpurun_out/t_full.log, smoke.log
cd /root/repo
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/t_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
