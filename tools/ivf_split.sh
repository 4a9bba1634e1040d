#!/bin/bash
# per-kernel times of one IVF search leg under rocprofv3 --kernel-trace: tools/ivf_split.sh <tag> <script args...>
# e.g. tools/ivf_split.sh new5m tools/perf_ivf5m.py 5000000 ; MRAG_IVFS_SCAN=128 tools/ivf_split.sh old5m tools/perf_ivf5m.py 5000000
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
rm -rf gpurun_out/split_$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/split_$tag -o t -- python3 "$@" > gpurun_out/split_$tag.log 2>&1
python3 - "$tag" <<PY
import csv,glob,sys,collections
f=glob.glob("gpurun_out/split_%s/**/*kernel_trace.csv" % sys.argv[1], recursive=True)[0]
by=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "ivf" in n or "bf_" in n: by[(n.split("(")[0].replace("void ","").replace("mrag::","")[:44], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print(sys.argv[1])
for k,v in sorted(by.items(), key=lambda kv:-sum(kv[1]))[:12]:
    v=sorted(v); print("   %-46s grid %-9s calls %3d  median %9.1f us  min %9.1f" % (k[0], k[1], len(v), v[len(v)//2], v[0]))
PY
