#!/bin/bash
# round-4 GPU session BP: k_c1_imgstats on eight workgroups per CU -- conv1 / sgan tests, step time, kernel time
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4bp; mkdir -p $O
R=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nn_gpu.py -x -q -k "conv1 or sgan" 2>&1 | tail -n 2
for k in 1 1; do timeout 600 python tools/bench_nn.py sgan --steps 100 2>&1 | tail -n 1 | cut -c60-200; done
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/tools/bench_nn.py sgan --steps 40 > $O/run.log 2>&1
python $R/tools/prof_summary.py stats $(find $O/prof -name "*.db" | head -1) > $O/sgan_stats.txt 2>&1; grep -E "k_c1_|k_sum_partials|k_bn_" $O/sgan_stats.txt | cut -c1-50,95-150
rm -rf $O/prof
