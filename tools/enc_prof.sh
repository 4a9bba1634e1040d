#!/bin/bash
# per-kernel times of one encoder shape: tools/enc_prof.sh minilm-l6 4096 128
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/encp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/encp -o e -- python3 -c "
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import perf_ivf_encoder as p
p.encoder('$1', $2, $3, iters=3)
" > gpurun_out/encp.log 2>&1
grep encoder gpurun_out/encp.log
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/encp/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("   %-70s calls %4s avg %9.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
