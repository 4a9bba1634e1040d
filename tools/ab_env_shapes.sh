#!/bin/bash
# same-box A/B of environments over K2 shapes (interleaved rounds): SHAPES="..." tools/ab_env_shapes.sh "MRAG_K2_TAILS=0" "MRAG_K2_TAILS=1"
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  for cfg in "$@"; do
    echo "== [$cfg]"
    env $cfg ITERS=${ITERS:-12} python tools/quick_perf.py ${SHAPES:-10000x1000000x768} 2>&1 | grep nq=
  done
done
