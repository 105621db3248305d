#!/bin/bash
# A variant of libttx.so that recompiles only the named translation units with extra -D flags and links the rest from the
# main build's objects (build/obj, python __graft_entry__.py):
#   scripts/build_variant_fast.sh <name> "<unit> [<unit> ..]" [-DFLAG=..]...   ->  fbtt-embedding_amd/variants/libttx_<name>.so
# e.g. scripts/build_variant_fast.sh np1 "ttx_tt_spec32 ttx_tt" -DTTX_NP32=1
set -e
NAME=$1; UNITS=$2; shift 2
cd "$(dirname "$0")/.."
mkdir -p fbtt-embedding_amd/variants build/obj_$NAME
cp build/obj/*.o build/obj_$NAME/
for u in $UNITS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=16 -Iinclude "$@" \
    -c fbtt-embedding_amd/csrc/$u.hip -o build/obj_$NAME/$u.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o fbtt-embedding_amd/variants/libttx_$NAME.so build/obj_$NAME/*.o
echo built fbtt-embedding_amd/variants/libttx_$NAME.so
