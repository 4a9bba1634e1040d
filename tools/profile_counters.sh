#!/bin/bash
# Counter evidence for every bench leg (north_star: "rocprof HBM GB/s and MFMA-busy counters"): separate rocprofv3 --pmc
# passes of the SAME bench command (all sub-objects on: north star / C2 / IVF / encoder kernels get rows too), each pass
# with --kernel-trace only (gpurun refuses --pmc together with the sys / hip / hsa trace domains).  The program comes
# directly after "--".  usage: tools/profile_counters.sh r03a   -> gpurun_out/<tag>_pmc_<pass>/ + <tag>_counters.json
set -e
tag=${1:-rXX}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=$PWD/gpurun_out
rocprofv3 -L > $o/${tag}_counter_list.txt 2>&1 || true
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $o/${tag}_pmc_$name -o ${tag} -- python3 bench.py --steps 3 --warmup 1 --no-cpu --skip-legs ivf_roofline_5m,ivf_skewed,ingest_from_text > $o/${tag}_pmc_$name.log 2>&1 \
    || { echo "pass $name failed"; tail -5 $o/${tag}_pmc_$name.log; }
}
pass busy   SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE
pass lds    SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA
pass l2     TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass fetch  FETCH_SIZE
pass write  WRITE_SIZE
python3 tools/pmc_counters_reduce.py $o/${tag}_counters.json $o/${tag}_pmc_busy $o/${tag}_pmc_lds $o/${tag}_pmc_l2 $o/${tag}_pmc_fetch $o/${tag}_pmc_write
