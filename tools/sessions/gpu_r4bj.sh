#!/bin/bash
# round-4 GPU session BJ: the failing conv1 case in full, with and without the packed kernel; packed-vs-general test
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_nn_gpu.py -q -k "fused_conv1" 2>&1 | grep -v "^$" | tail -n 40
echo "---- RML_C1_PK=0"
RML_C1_PK=0 timeout 600 python -m pytest tests/test_nn_gpu.py -q -k "fused_conv1" 2>&1 | tail -n 8
echo "---- packed vs general"
timeout 600 python -m pytest tests/test_nn_gpu.py -q -k "packed_conv1" 2>&1 | tail -n 12
