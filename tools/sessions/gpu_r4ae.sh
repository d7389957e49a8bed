#!/bin/bash
# round-4 GPU session AE: C-ABI tests of the CNN-chain entry points; batch size of the CNN chain
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_capi_gpu.py -x -q 2>&1 | tail -n 5
for rep in 1 2; do
  for bs in 8192 16384 32768; do
    printf "batch %5d: " $bs; timeout 300 python tools/dnn_chain.py --batch $bs
  done
done
