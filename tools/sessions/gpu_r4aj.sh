#!/bin/bash
# round-4 GPU session AJ: does the projection's rate depend on where the volumes live? (tools/exp/placement.py), three processes per grid
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_nn_gpu.py -x -q -k "dense" 2>&1 | tail -n 2
for rep in 1 2 3; do timeout 600 python tools/exp/placement.py --grid 22x31x176 --frames 16384; done
for rep in 1 2; do RML_WAVE_SHARE=1 timeout 600 python tools/exp/placement.py --grid 64x64x128 --frames 8192; done
