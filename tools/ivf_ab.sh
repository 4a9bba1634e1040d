#!/bin/bash
# same-box A/B of builds on the IVF legs (interleaved rounds): tools/ivf_ab.sh build_ab/a.so build_ab/b.so   (ROWS=5000000 for the whole corpus)
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  for lib in "$@"; do
    if [ -n "$ROWS" ]; then
      echo "$(basename $lib): $(MRAG_HIP_LIB=$PWD/$lib python tools/perf_ivf5m.py $ROWS 2>&1 | grep -v amdgpu | tail -1 | grep -o "kernel_ms.: [0-9.]*, .search_ms.: [0-9.]*")"
    else
      echo "$(basename $lib): $(MRAG_HIP_LIB=$PWD/$lib python tools/perf_ivf_encoder.py ivf 2>&1 | grep "list scan")"
    fi
  done
done
