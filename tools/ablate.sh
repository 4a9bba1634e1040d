#!/bin/bash
# K2 ablations on the GPU box (diagnostic build, timing only; results are wrong by design):
#   1 no epilogue | 4 no loads | 8 no LDS reads / MFMA | 32 no list pushes | 128 every corpus tile = the first (L2-resident)
set -e
cd "$(dirname "$0")/.."
make -C a-modular-rag-framework_amd/csrc DIAG=1 -B > gpurun_out/ablate_make.log 2>&1
for f in ${FLAGS:-0 1 5 9 137 129 128}; do
  echo "flags=$f"
  MRAG_DEBUG_FLAGS=$f python tools/quick_perf.py ${SHAPE:-10000x1000000x768}
done
