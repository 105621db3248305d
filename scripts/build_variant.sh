#!/bin/bash
# build an experimental variant of libttx.so: scripts/build_variant.sh <name> [-DFLAG=..]...
# -> variants/libttx_<name>.so ; run with TTX_LIB=variants/libttx_<name>.so python bench.py ...
set -e
NAME=$1; shift
mkdir -p variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function "$@" \
  -o variants/libttx_$NAME.so fbtt-embedding_amd/csrc/*.hip
echo built variants/libttx_$NAME.so
