#!/bin/bash
# build an experimental variant of libttx.so with other -D tuning macros (TTX_HOT_NF, TTX_SEG_THIN, TTX_PLAN_XWG, ..):
#   scripts/build_variant.sh <name> [-DFLAG=..]...   ->  fbtt-embedding_amd/variants/libttx_<name>.so
# A/B it on the GPU box with scripts/variants.sh (ctypes route, TTX_LIB) or TTX_LIB=... python bench.py --no-graph ...
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
mkdir -p fbtt-embedding_amd/variants
OBJ=build/obj_$NAME; mkdir -p $OBJ
ls fbtt-embedding_amd/csrc/*.hip | xargs -P 8 -I{} sh -c '/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=16 -Iinclude "$@" -c {} -o '$OBJ'/$(basename {} .hip).o' _ "$@"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o fbtt-embedding_amd/variants/libttx_$NAME.so $OBJ/*.o
echo built fbtt-embedding_amd/variants/libttx_$NAME.so
