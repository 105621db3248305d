#!/bin/bash
# build an experimental variant of libttx.so with other -D tuning macros (TTX_HOT_NF, TTX_SEG_THIN, TTX_PLAN_XWG, ..):
#   scripts/build_variant.sh <name> [-DFLAG=..]...   ->  fbtt-embedding_amd/variants/libttx_<name>.so
# A/B it on the GPU box with scripts/variants.sh (ctypes route, TTX_LIB) or TTX_LIB=... python bench.py --no-graph ...
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
mkdir -p fbtt-embedding_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Wno-pass-failed -Iinclude "$@" \
  -o fbtt-embedding_amd/variants/libttx_$NAME.so fbtt-embedding_amd/csrc/ttx_api.hip fbtt-embedding_amd/csrc/ttx_cache.hip \
  fbtt-embedding_amd/csrc/ttx_plan.hip fbtt-embedding_amd/csrc/ttx_tt.hip
echo built fbtt-embedding_amd/variants/libttx_$NAME.so
