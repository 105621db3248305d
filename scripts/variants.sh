#!/bin/bash
# A/B timing of kernel build variants: fbtt-embedding_amd/variants/libttx_<v>.so (built with other -D tuning macros),
# selected with TTX_LIB through the ctypes route.  usage (GPU box): scripts/variants.sh <tag> "<workloads>" <v1> <v2> ...
set -u
TAG=$1; WLS=$2; shift 2
for V in "$@"; do
  echo "#### variant $V"
  TTX_LIB=$(pwd)/fbtt-embedding_amd/variants/libttx_$V.so TTX_NO_NATIVE_NODE=1 scripts/kprof.sh ${TAG}_$V $WLS 2>&1 | grep -E "^##|ttx::"
done
