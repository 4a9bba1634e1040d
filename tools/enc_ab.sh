#!/bin/bash
# same-box A/B of encoder builds (interleaved rounds): tools/enc_ab.sh build_ab/base.so build_ab/roles.so
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  for lib in "$@"; do
    echo "$(basename $lib): $(MRAG_HIP_LIB=$PWD/$lib python tools/perf_ivf_encoder.py ${WHAT:-enc} 2>&1 | grep encoder | tr '\n' ' ')"
  done
done
