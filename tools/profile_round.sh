#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the bench command, then FETCH_SIZE / WRITE_SIZE in
# their own --pmc passes.  The trace pass runs every bench sub-object (north star, C2, IVF, encoder), so its
# kernel_stats.csv also holds the IVF list-scan and encoder kernels.  usage: tools/profile_round.sh r01e   (outputs under gpurun_out/<tag>_*)
set -e
tag=${1:-rXX}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=$PWD/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_trace -o ${tag} -- python3 bench.py --steps 5 --warmup 2 --no-cpu > $o/${tag}_bench_trace.log 2>&1
find $o/${tag}_trace -name "*kernel_stats.csv" -exec cp {} $o/${tag}_bench_n1_kernel_stats.csv \;
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/${tag}_pmc_fetch -o ${tag} -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras > $o/${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/${tag}_pmc_write -o ${tag} -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras > $o/${tag}_pmc_write.log 2>&1
python3 tools/pmc_reduce.py $o/${tag}_pmc_traffic.json $o/${tag}_pmc_fetch $o/${tag}_pmc_write
head -8 $o/${tag}_bench_n1_kernel_stats.csv
