#!/bin/bash
# per-kernel times of the score-segment IVF regime for the timing-only ablation builds build_ab/ivf_d<bits>.so
# (csrc/ivf_scan.hip: MRAG_IVFS_DIAG bits 1 no stores | 2 no MFMA | 4 no query gather | 8 no corpus loads)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for d in ${DIAGS:-0 1 2 4 8 15}; do
  rm -rf gpurun_out/ivfa
  MRAG_HIP_LIB=$PWD/build_ab/ivf_d$d.so rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ivfa -o t -- python3 tools/perf_ivf_encoder.py ivf > gpurun_out/ivfa.log 2>&1
  python3 - "$d" <<PY
import csv,glob,sys,collections
f=glob.glob("gpurun_out/ivfa/**/*kernel_trace.csv", recursive=True)[0]
by=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "ivfs" in n: by[(n[:40], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("diag", sys.argv[1], " ".join("%s[%s] %.0f us;" % (k[0].split("::")[1][:18], k[1], sorted(v)[len(v)//2]) for k,v in sorted(by.items())))
PY
done
