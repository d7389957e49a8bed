#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh rNN
# 1. rocprofv3 --kernel-trace --stats of the default bench workload (fewer steps, no CPU leg)
# 2. two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel trace only -- never with sys/hip/hsa traces)
# 3. kernel stats of the dnn pipeline (tools/bench_nn.py dnn) and of the sgan train step (tools/bench_nn.py sgan)
# Summaries land in gpurun_out/; copy the ones to keep into profiles/.
set -u
TAG=${1:-rXX}
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
BENCH="python $R/bench.py --no-cpu --parity 512 --steps 5 --warmup 2"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o k -- $BENCH > $R/gpurun_out/${TAG}_bench_under_rocprofv3.json 2> $R/gpurun_out/prof_stats.err
python $R/tools/prof_summary.py stats $R/gpurun_out/prof_stats/k_results.db > $R/gpurun_out/${TAG}_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/prof_$c -o k -- $BENCH --no-walabot > /dev/null 2> $R/gpurun_out/prof_$c.err
done
cd $R
python tools/prof_summary.py pmc gpurun_out/prof_FETCH_SIZE/k_results.db gpurun_out/prof_WRITE_SIZE/k_results.db > gpurun_out/${TAG}_pmc.txt
cp profiles/pmc_latest.json gpurun_out/pmc_latest.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_nn -o k -- python $R/tools/bench_nn.py dnn --frames 16384 > $R/gpurun_out/${TAG}_bench_nn_dnn.json 2> $R/gpurun_out/prof_nn.err
python $R/tools/prof_summary.py stats $R/gpurun_out/prof_nn/k_results.db > $R/gpurun_out/${TAG}_kernel_stats_dnn.txt
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sgan -o k -- python $R/tools/bench_nn.py sgan --steps 100 > $R/gpurun_out/${TAG}_bench_nn_sgan.json 2> $R/gpurun_out/prof_sgan.err
python $R/tools/prof_summary.py stats $R/gpurun_out/prof_sgan/k_results.db > $R/gpurun_out/${TAG}_kernel_stats_sgan.txt
cd $R
rm -rf gpurun_out/prof_stats gpurun_out/prof_FETCH_SIZE gpurun_out/prof_WRITE_SIZE gpurun_out/prof_nn gpurun_out/prof_sgan
head -12 gpurun_out/${TAG}_kernel_stats.txt | cut -c1-160
head -12 gpurun_out/${TAG}_pmc.txt | cut -c1-170
head -8 gpurun_out/${TAG}_kernel_stats_dnn.txt | cut -c1-160
