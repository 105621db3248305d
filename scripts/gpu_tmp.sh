#!/bin/bash
mkdir -p gpurun_out
timeout 300 python scripts/host_time.py 2>&1 | tail -8
