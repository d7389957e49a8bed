#!/bin/bash
# round-4 GPU session AK: placement of the volumes / of the code rows against the projection's rate (tools/exp/placement.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do timeout 600 python tools/exp/placement.py --grid 22x31x176 --frames 16384 | tail -n 8; done
RML_WAVE_SHARE=1 timeout 600 python tools/exp/placement.py --grid 64x64x128 --frames 8192 | tail -n 8
