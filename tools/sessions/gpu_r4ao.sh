#!/bin/bash
# round-4 GPU session AO: k_pre3 projection by projection (intermediate of one projection: four workgroups per CU) -- tests, kernel alone, chain
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_nn_gpu.py -x -q -k "preprocess or kblock or dense or predict_volumes or dnn" 2>&1 | tail -n 4
for rep in 1 2; do timeout 300 python tools/pre3_bench.py; timeout 300 python tools/dnn_chain.py; done
