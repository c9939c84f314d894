#!/bin/bash
# Experiment aid: build classeq2_amd/csrc/libclsplace_<tag>.so with extra -D flags for cls_kernels.hip (the other objects come
# from the normal build).  A tool picks it up through CLS_PLACE_LIB=<path> (classeq2_amd/engine.py).
# usage: bash tools/build_variant.sh <tag> -DFAST_MIN_WAVES=8 -DCLS_NARROW_CANON_BITS=8 ...
set -e
TAG=$1; shift
cd "$(dirname "$0")/../classeq2_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -fno-strict-aliasing -DMIN_WAVES_PER_EU=1 -I../../include "$@" -c cls_kernels.hip -o _obj/cls_kernels_$TAG.o
OBJS=$(ls _obj/*.o | grep -v "cls_kernels")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -pthread _obj/cls_kernels_$TAG.o $OBJS -o libclsplace_$TAG.so
echo built libclsplace_$TAG.so
