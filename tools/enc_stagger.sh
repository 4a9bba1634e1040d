#!/bin/bash
# per-kernel encoder GEMM times for several XCD start staggers (MRAG_ENC_STAGGER cycles per XCD; -1 = library default)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for s in ${STAGGERS:-0 -250 -500 -1000 -2000}; do
  rm -rf gpurun_out/enc_stg
  MRAG_ENC_STAGGER=$s rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/enc_stg -o enc -- python3 tools/perf_ivf_encoder.py enc-bge > gpurun_out/enc_stg.log 2>&1
  echo "== stagger=$s: $(grep encoder gpurun_out/enc_stg.log)"
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/enc_stg/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print("   %-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
