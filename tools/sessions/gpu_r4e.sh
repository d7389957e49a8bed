#!/bin/bash
# round-4 GPU session E: k_derive_slice with one rolling register buffer (slot refilled behind its reduction), s_r staged in halves,
# gather batches of 8; U = one or two periods in flight
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4e; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_projection_gpu.py -x -q -k "derive or slice" 2>&1 | tail -n 5
for u2 in 0 1; do for g in 64x64x128 22x31x176; do
  for pc in 2 3 4 5; do
    RML_DERIVE_U2=$u2 RML_DERIVE_PERCU=$pc timeout 300 python tools/kbench.py derive --grid $g --frames 16384 --iters 8 2>&1 | grep "k_derive_slice" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   u2=$u2 $g percu $pc', d['what'][:40], d['ms_med'], d['frac_of_8TBs'])"
  done
done; done
