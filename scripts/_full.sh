cd /root/repo
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/t_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
