#!/bin/bash
# K2 ablations on the GPU box (diagnostic builds, timing only; results are wrong by design).
# Flags are compile-time (-DMRAG_DIAG=<flags>), one rebuild of bf_index.hip per flag set:
#   1 no epilogue | 4 no loads | 8 no LDS reads / MFMA | 32 no list pushes | 256 in-kernel clock
set -e
cd "$(dirname "$0")/.."
for f in ${FLAGS:-1 5 9}; do
  touch a-modular-rag-framework_amd/csrc/bf_index.hip
  make -C a-modular-rag-framework_amd/csrc DIAG=$f > gpurun_out/ablate_make.log 2>&1
  for pp in ${PIPES:-0 1}; do
    echo "flags=$f pipe=$pp"
    MRAG_PIPE=$pp ITERS=${ITERS:-5} python tools/quick_perf.py ${SHAPE:-10000x1000000x768} 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-1}
  done
done
