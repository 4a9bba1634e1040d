#!/bin/bash
# same-box A/B of two builds of libmrag_hip.so (interleaved rounds): tools/ab.sh build_ab/base.so build_ab/new.so
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  for lib in "$@"; do
    echo -n "$(basename $lib): "
    MRAG_HIP_LIB=$PWD/$lib ITERS=${ITERS:-20} python tools/quick_perf.py ${SHAPE:-10000x1000000x768} 2>&1 | grep -v amdgpu.ids
  done
done
