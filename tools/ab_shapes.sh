#!/bin/bash
# same-box A/B of builds of libmrag_hip.so over several shapes in one process each (interleaved rounds):
#   SHAPES="10000x1000000x768 1000x1000000x768 1000x100000x768" tools/ab_shapes.sh build_ab/base.so build_ab/new.so
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  for lib in "$@"; do
    echo "== $(basename $lib)"
    MRAG_HIP_LIB=$PWD/$lib ITERS=${ITERS:-12} python tools/quick_perf.py ${SHAPES:-10000x1000000x768 1000x1000000x768 1000x100000x768} 2>&1 | grep -v amdgpu.ids
  done
done
