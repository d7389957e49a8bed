#!/bin/bash
# round-4 GPU session A: first run of k_slice_rows / k_derive_slice -- parity tests, then kernel-level rates
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_projection_gpu.py -x -q -k "derive or slice or index" 2>&1 | tail -n 25
timeout 600 python -m pytest tests/test_svm_gpu.py -x -q -k "slice or nan_row" 2>&1 | tail -n 8
for g in 64x64x128 22x31x176; do
  for w in derive slice; do
    timeout 300 python tools/kbench.py $w --grid $g --frames 16384 --iters 10 2>&1 | tee -a $O/kbench.txt | cut -c1-260
  done
done
timeout 300 python tools/kbench.py derive --grid 64x64x128 --frames 16384 --iters 10 --u8 2>&1 | tee -a $O/kbench.txt | cut -c1-260
timeout 300 python tools/kbench.py derive --grid 22x31x176 --frames 32768 --iters 10 --u8 2>&1 | tee -a $O/kbench.txt | cut -c1-260
