#!/bin/bash
# Encoder GEMM ablations on the GPU box (timing only): per-kernel times of one bge-base forward per diag build.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for f in ${FLAGS:-0 1 2 4 8}; do
  touch a-modular-rag-framework_amd/csrc/encoder.hip
  make -C a-modular-rag-framework_amd/csrc ENCDIAG=$f > gpurun_out/enc_ablate_make.log 2>&1
  rm -rf gpurun_out/enc_abl_$f
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/enc_abl_$f -o enc -- python3 tools/perf_ivf_encoder.py enc-bge > gpurun_out/enc_abl_$f.log 2>&1
  echo "== ENCDIAG=$f: $(grep encoder gpurun_out/enc_abl_$f.log)"
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/enc_abl_$f/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print("   %-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
