#!/bin/bash
# ring-of-stages main-loop experiment (csrc/experiments/k2r_proto.hip): every built variant at the headline and north-star shapes
cd "$(dirname "$0")/.."
for sh in "10000 1000000 768 10" "1000 1000000 768 20"; do
  for r in 1 2; do
    for b in build_ab/k2r_*; do $b $sh; done
  done
done
