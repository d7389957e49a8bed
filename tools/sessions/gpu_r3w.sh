#!/bin/bash
# round-3 GPU session W: MIOpen solver choice for the SGAN convolutions (heuristic pick vs timed find) 
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3w; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for b in 0 1; do
  echo "== cudnn.benchmark=$b"
  RML_CUDNN_BENCHMARK=$b timeout 900 python tools/bench_nn.py sgan --steps 100 2>&1 | grep -v amdgpu.ids | cut -c1-300
done
cd /tmp
RML_CUDNN_BENCHMARK=1 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o k -- python $R/tools/bench_nn.py sgan --steps 100 > $R/$O/sgan_b1.json 2> $R/$O/prof.err
cd $R
python tools/prof_summary.py stats $O/prof/k_results.db > $O/stats_sgan_benchmark1.txt; rm -rf $O/prof
head -n 24 $O/stats_sgan_benchmark1.txt | cut -c1-170
