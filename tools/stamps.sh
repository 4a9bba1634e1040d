#!/bin/bash
# s_memtime stamps of workgroup 0, tiles 60..65 of its split (diagnostic build DIAG=16)
set -e
cd "$(dirname "$0")/.."
touch a-modular-rag-framework_amd/csrc/bf_index.hip
make -C a-modular-rag-framework_amd/csrc DIAG=16 > gpurun_out/stamps_make.log 2>&1
ITERS=3 python tools/quick_perf.py ${SHAPE:-10000x1000000x768} 2>&1 | grep -v amdgpu.ids | tail -4
