#!/bin/bash
# round-4 GPU session U: fused CNN preprocessing + fused dense tail -- tests, chain A/B, kernel summary
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4u; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_nn_gpu.py -x -q -k "preprocess or predict_volumes or dnn or dense" 2>&1 | tail -n 15
for rep in 1 2; do
  timeout 300 python tools/dnn_chain.py --exact
  timeout 300 python tools/dnn_chain.py
  timeout 300 python tools/dnn_chain.py --u8
done
R=$PWD
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o dnn -- python $R/tools/dnn_chain.py --steps 3 > $R/$O/prof.log 2>&1; cd $R
python tools/prof_summary.py stats $O/prof/dnn_results.db > $O/stats_dnn_chain.txt; head -16 $O/stats_dnn_chain.txt | cut -c1-150
rm -rf $O/prof
