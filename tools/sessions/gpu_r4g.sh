#!/bin/bash
# round-4 GPU session G: the whole GPU suite, then gather batch size of k_derive_slice (4 / 8 / 16 quads) on balanced launches
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4g; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -n 12
for v in "" ub4 ub16; do for g in 64x64x128 22x31x176; do
  if [ -n "$v" ]; then export RML_LIB=$PWD/radar-ml_amd/libradarml_hip_$v.so; else unset RML_LIB; fi
  timeout 300 python tools/kbench.py derive --grid $g --frames 24576 --iters 8 2>&1 | grep "k_derive_slice" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   variant=$v $g', d['what'][:40], d['ms_med'], d['frac_of_8TBs'])"
done; done
