#!/bin/bash
# round-4 GPU session B: what bounds k_derive_slice (ablation builds: 1 = no LDS atomics, 2 = no tail masks, 4 = no s_r accumulators)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4b; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_svm_gpu.py -x -q -k "nan_row" 2>&1 | grep -v "^E    *\[" | tail -n 30
for v in "" dsabl1 dsabl2 dsabl4 dsabl7; do
  for g in 64x64x128 22x31x176; do
    echo "== variant '$v' grid $g"
    if [ -n "$v" ]; then export RML_LIB=$PWD/radar-ml_amd/libradarml_hip_$v.so; else unset RML_LIB; fi
    for pc in 1 2 3 4; do
      RML_DERIVE_PERCU=$pc timeout 300 python tools/kbench.py derive --grid $g --frames 16384 --iters 8 2>&1 | grep "k_derive_slice" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   percu $pc', d['what'][:40], d['ms_med'], d['frac_of_8TBs'])"
    done
  done
done
