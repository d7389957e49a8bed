#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_nn_gpu.py -x -q -k "dnn or resize or preprocess" 2>&1 | grep -v "^E    *tensor\|device='cuda" | tail -n 30
