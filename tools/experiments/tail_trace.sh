cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tailtr
MRAG_K2_TAILS=1 ITERS=3 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tailtr -o t -- python3 tools/quick_perf.py 10000x1000000x768 > gpurun_out/tailtr.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/tailtr/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "bf_gemm_topk" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
for r in rows[-6:]:
    print(r.get("Queue_Id"), r.get("Grid_Size_X", r.get("Grid_Size")), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3, "us dur", (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
PY
rm -rf gpurun_out/tailtr
