#!/bin/bash
# per-kernel times of one bge-base forward for a given build: tools/enc_prof_lib.sh build_ab/roles.so [arch B S]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
lib=$1; arch=${2:-bge-base}; B=${3:-2048}; S=${4:-128}
tag=$(basename $lib .so)
rm -rf gpurun_out/encp_$tag
MRAG_HIP_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/encp_$tag -o e -- python3 -c "
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import perf_ivf_encoder as p
p.encoder('$arch', $B, $S, iters=3)
" > gpurun_out/encp_$tag.log 2>&1
echo "== $tag: $(grep encoder gpurun_out/encp_$tag.log)"
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/encp_$tag/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print("   %-64s calls %4s avg %9.1f us  %5s%%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
rm -rf gpurun_out/encp_$tag
