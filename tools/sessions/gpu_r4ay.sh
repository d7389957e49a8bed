#!/bin/bash
# round-4 GPU session AY: where the SGAN step's wall time goes -- kernel trace of 40 steps, busy / idle / gaps by following kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4ay; mkdir -p $O
R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 600 python $R/tools/bench_nn.py sgan --steps 100 2>&1 | tail -n 2
timeout 900 rocprofv3 --kernel-trace -d $O/prof -o k -- python $R/tools/bench_nn.py sgan --steps 40 > $O/run.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
python $R/tools/step_gaps.py $DB --marker k_c1_imgstats --per-step 6 --steps 10 --dump 520 > $O/sgan_gaps.txt 2>&1
head -100 $O/sgan_gaps.txt
rm -rf $O/prof
