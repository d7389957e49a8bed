#!/bin/bash
# round-4 GPU session C: k_derive_slice with the b128 read-add-write strip (no LDS atomics); request granularity of the xy gather
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4c; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_projection_gpu.py -x -q -k "derive or slice" 2>&1 | tail -n 5
timeout 600 python -m pytest tests/test_svm_gpu.py -x -q -k "slice or nan_row" 2>&1 | tail -n 3
for g in 64x64x128 22x31x176; do
  for pc in 1 2 3 4; do
    RML_DERIVE_PERCU=$pc timeout 300 python tools/kbench.py derive --grid $g --frames 16384 --iters 8 2>&1 | grep "k_derive_slice" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   $g percu $pc', d['what'][:40], d['ms_med'], d['frac_of_8TBs'])"
  done
done
for g in 64x64x128 22x31x176; do
  timeout 600 python tools/pmc_kernels.py --match 'k_slice_rows|k_derive_slice|k_project_slice' --counters FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -- python tools/kbench.py slice --grid $g --frames 16384 --iters 3 2>&1 | tee -a $O/pmc_slice.txt | cut -c1-400
  timeout 600 python tools/pmc_kernels.py --match 'k_slice_rows|k_derive_slice|k_project_slice' --counters FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -- python tools/kbench.py derive --grid $g --frames 16384 --iters 3 2>&1 | tee -a $O/pmc_derive.txt | cut -c1-400
done
