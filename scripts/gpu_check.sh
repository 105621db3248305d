#!/bin/bash
# GPU box, repo root: the -m gpu suite, then rocprofv3 per-kernel durations of the named workloads (scripts/kprof.sh).
#   scripts/gpu_check.sh <tag> [workload ...]     -> gpurun_out/t_<tag>.log, gpurun_out/kprof_<tag>/*.md
TAG=$1; shift
mkdir -p gpurun_out
if [ -z "${SKIP_TESTS:-}" ]; then
  timeout 900 python -m pytest tests -x -q -m gpu ${PYTEST_ARGS:-} 2>&1 | tail -15 > gpurun_out/t_$TAG.log
  cat gpurun_out/t_$TAG.log
fi
[ $# -gt 0 ] && timeout 1500 scripts/kprof.sh $TAG "$@" 2>&1 | grep -v "^$"
