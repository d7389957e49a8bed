#!/bin/bash
# round-3 GPU session AL2: k_c1_bwd1 with the row's gradient loads issued up front, 4 (default) / 8 channels per thread
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3al; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nn_gpu.py -x -q 2>&1 | tail -n 3
for rep in 1 2; do for c in 4 8; do
  echo "== CPT=$c"; RML_C1_CPT=$c timeout 600 python tools/bench_nn.py sgan --steps 100 2>&1 | grep configs | cut -c1-200
done; done
R=$PWD; cd /tmp
for c in 4 8; do
  RML_C1_CPT=$c timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof$c -o k -- python $R/tools/bench_nn.py sgan --steps 40 > /dev/null 2> $R/$O/prof$c.err
  python $R/tools/prof_summary.py stats $R/$O/prof$c/k_results.db | grep "k_c1_bwd1" | cut -c1-170; rm -rf $R/$O/prof$c
done
