#!/bin/bash
# In-kernel stamps of the FUSED-kernel IVF list scan (descriptor mode, forced with MRAG_IVF_SCORES_MB=0), workgroup 0
# (diagnostic build DIAG=16; timing shares only, see DESIGN).  The score-segment scan has its own: ivf_stamps2.py.
set -e
cd "$(dirname "$0")/.."
# (the diagnostic library goes to build_ab/, selected with MRAG_HIP_LIB: the in-tree product library stays the production build)
tools/build_variant.sh ivf_diag16 -DMRAG_DIAG=16 > gpurun_out/ivf_stamps_make.log 2>&1
MRAG_HIP_LIB=$PWD/build_ab/ivf_diag16.so MRAG_IVF_SCORES_MB=0 python tools/perf_ivf_encoder.py ivf 2>&1 | grep -v amdgpu.ids | grep -E "stamps|tail|IVF" | tail -${TAIL:-8}
