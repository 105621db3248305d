#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/t_b9.log
scripts/kprof.sh b9 cfg5shard tb4 cfg2 | grep -E "^##|spec_|eager"
echo "#### nobfs variant"
TTX_LIB=$(pwd)/fbtt-embedding_amd/variants/libttx_nobfs.so TTX_NO_NATIVE_NODE=1 scripts/kprof.sh b9n cfg5shard tb4 cfg2 | grep -E "^##|spec_|eager"
echo "#### main, ctypes route"
TTX_NO_NATIVE_NODE=1 scripts/kprof.sh b9m cfg5shard | grep -E "^##|spec_|eager"
