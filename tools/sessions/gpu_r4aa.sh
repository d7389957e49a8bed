#!/bin/bash
# round-4 GPU session AA: whole GPU suite on the tree with the fused CNN chain + code stage; dnn chain numbers
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4aa; mkdir -p $O
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -n 6 $O/pytest.log
for rep in 1 2; do
  timeout 300 python tools/dnn_chain.py
  timeout 300 python tools/dnn_chain.py --u8
done
R=$PWD
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o dnn -- python $R/tools/dnn_chain.py --steps 3 > $R/$O/prof.log 2>&1; cd $R
python tools/prof_summary.py stats $O/prof/dnn_results.db > $O/stats_dnn_chain.txt; head -12 $O/stats_dnn_chain.txt | cut -c1-150
rm -rf $O/prof
