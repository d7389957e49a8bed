#!/bin/bash
# round-4 GPU session BA: k_c1_bwd1_pk (packed first-layer backward) -- tests, step time against RML_C1_PK=0, kernel times
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4ba; mkdir -p $O
R=$PWD
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_nn_gpu.py -x -q -k "sgan or bn_lrelu or conv1" 2>&1 | tail -n 4
for k in 0 1 0 1; do echo "RML_C1_PK=$k"; RML_C1_PK=$k timeout 600 python tools/bench_nn.py sgan --steps 100 2>&1 | tail -n 1; done
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/tools/bench_nn.py sgan --steps 40 > $O/run.log 2>&1
python $R/tools/prof_summary.py $(find $O/prof -name "*.db" | head -1) 2>/dev/null | grep -E "k_c1_|k_bn_|k_sum" | cut -c1-190
rm -rf $O/prof
