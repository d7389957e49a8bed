#!/bin/bash
# round-4 GPU session AV: code rows of the pipeline's workspaces written read-compare-write (RML_CODE_RMW=1: only what changed is stored) -- tests, A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4av; mkdir -p $O
export TMPDIR=/tmp
RML_CODE_RMW=1 timeout 1200 python -m pytest tests/test_svm_gpu.py -x -q 2>&1 | tail -n 3
B="python bench.py --steps 8 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 1024"
for rep in 1 2 3; do for k in 0 1; do
  RML_CODE_RMW=$k timeout 900 $B > $O/rmw${k}_$rep.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/rmw${k}_$rep.json rmw$k | grep -v "gate\|slice_mode" | cut -c1-125
done; done
