#!/bin/bash
# round-4 GPU session AL: frames of a wave side by side (w, w + #waves, ...) or as one contiguous slab (RML_SLAB=1) against the placement of the volumes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_projection_gpu.py -x -q -k "code_stage or linear_plane" 2>&1 | tail -n 2
RML_SLAB=1 timeout 600 python -m pytest tests/test_projection_gpu.py -x -q -k "code_stage or linear_plane" 2>&1 | tail -n 2
for rep in 1 2; do for sl in 0 1; do echo "RML_SLAB=$sl"; RML_SLAB=$sl timeout 600 python tools/exp/placement.py --grid 22x31x176 --frames 16384 | head -n 6; done; done
