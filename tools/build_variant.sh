#!/bin/bash
# Experiment aid: build classeq2_amd/csrc/libclsplace_<tag>.so with extra -D flags for ONE kernel file (default
# cls_kernels.hip; SRC=cls_tile.hip for the long-read kernel); the other objects come from the normal build, which this
# script runs first so that they are current.  A tool picks the library up through CLS_PLACE_LIB=<path> (classeq2_amd/engine.py).
# usage: [SRC=cls_tile.hip] bash tools/build_variant.sh <tag> -DFAST_MIN_WAVES=8 -DCLS_NARROW_CANON_BITS=8 ...
set -e
TAG=$1; shift
SRC=${SRC:-cls_kernels.hip}
BASE=${SRC%.*}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
(cd "$ROOT" && python -c 'import __graft_entry__ as g; g.build(only_missing=True)' >/dev/null)
cd "$ROOT/classeq2_amd/csrc"
mkdir -p _var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -fno-strict-aliasing -DMIN_WAVES_PER_EU=1 -I../../include "$@" -c $SRC -o _var/${BASE}_$TAG.o
OBJS=$(ls _obj/*.o | grep -v "_obj/${BASE}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -pthread _var/${BASE}_$TAG.o $OBJS -o libclsplace_$TAG.so
echo built libclsplace_$TAG.so
