cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_ivf.py tests/test_gpu_fullsize.py -x -q -m gpu -k "ivf or c5 or C5" 2>&1 | tail -3 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ivfp -o ivf -- python3 tools/perf_ivf_encoder.py ivf > gpurun_out/ivfp.log 2>&1
grep -E "scan|IVF" gpurun_out/ivfp.log
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/ivfp/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("   %-70s calls %4s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
