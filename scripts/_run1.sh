cd /root/repo
python -m pytest tests/test_dedup_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/t_dedup.log
for W in cfg5z-dedup tb4z-dedup; do python bench.py --workload $W --steps 30 --repeats 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/b_$W.json; done
scripts/kprof.sh gs cfg5z-dedup tb4z-dedup > gpurun_out/kprof_gs.md 2>&1
