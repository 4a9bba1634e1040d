#!/bin/bash
# list-scan ring depth sweep of the score-segment IVF regime (MRAG_IVFS_STAGES = 2 | 3 | 4)
cd "$(dirname "$0")/.."
for s in ${STAGES:-2 3 4}; do
  echo "== stages=$s"
  MRAG_IVFS_STAGES=$s python3 tools/perf_ivf_encoder.py ivf 2>&1 | grep -E "scan|IVF"
done
