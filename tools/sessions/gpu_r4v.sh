#!/bin/bash
# round-4 GPU session V: K-block feature layout between trunk and dense tail; rolling-refill k_project_lin A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4v; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_nn_gpu.py -x -q -k "preprocess or predict_volumes or dnn or dense or kblock" 2>&1 | tail -n 8
timeout 600 python -m pytest tests/test_projection_gpu.py -x -q -k "linear_plane or lin" 2>&1 | tail -n 3
for rep in 1 2; do
  for roll in 0 1; do
    echo "RML_LIN_ROLL=$roll"
    RML_LIN_ROLL=$roll timeout 300 python tools/kbench.py proj --grid 22x31x176 --frames 16384 | grep "codes+stats only" | cut -c1-220
    RML_LIN_ROLL=$roll timeout 300 python tools/dnn_chain.py
  done
done
R=$PWD
cd /tmp && RML_LIN_ROLL=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o dnn -- python $R/tools/dnn_chain.py --steps 3 > $R/$O/prof.log 2>&1; cd $R
python tools/prof_summary.py stats $O/prof/dnn_results.db > $O/stats_dnn_chain.txt; head -12 $O/stats_dnn_chain.txt | cut -c1-150
rm -rf $O/prof
