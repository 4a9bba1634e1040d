#!/bin/bash
# same-box A/B of encoder runs under different environments: tools/enc_ab_env.sh "LIB=build_ab/a.so X=1" "LIB=build_ab/b.so"
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  for cfg in "$@"; do
    echo "[$cfg] $(env $cfg bash -c 'MRAG_HIP_LIB=$PWD/$LIB python tools/perf_ivf_encoder.py ${WHAT:-enc-bge}' 2>&1 | grep encoder | tr '\n' ' ')"
  done
done
