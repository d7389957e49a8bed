#!/bin/bash
# round-4 GPU session AM: property test of the fused preprocessing on random geometries; dnn tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_nn_gpu.py -x -q -k "preprocess or kblock or dense or predict_volumes" 2>&1 | tail -n 12
