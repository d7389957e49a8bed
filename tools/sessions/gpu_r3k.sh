#!/bin/bash
# round-3 GPU session K: the 128x128 ring GEMM beside the projection (Walabot grid: k_project_lin BALLAST variant)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3k; mkdir -p $O
timeout 900 python -m pytest tests/test_svm_gpu.py -x -q -k "big or 256 or tile" 2>&1 | tail -4 > $O/pytest_svm.txt
timeout 900 python -m pytest tests/test_svm_gpu.py -x -q 2>&1 | tail -4 >> $O/pytest_svm.txt
B="python bench.py --grid 22x31x176 --frames 262144 --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 1024"
for rep in 1 2; do
  for r in 0 1; do
    R=""; [ $r = 1 ] && R=1
    RML_GEMM_RING128=$R timeout 900 $B > $O/wal_r128_${r}_$rep.json 2>> $O/wal.err
  done
done
# alone: the two 128x128 kernels on code rows
for r in 0 1; do
  R=""; [ $r = 1 ] && R=1
  RML_GEMM_RING128=$R RML_GEMM_BIG=0 timeout 300 python tools/kbench.py gemm --grid 22x31x176 --frames 8192 --svs 2281 --iters 20 2>/dev/null | tail -1 | cut -c1-200 >> $O/alone.txt
  RML_GEMM_RING128=$R RML_GEMM_BIG=0 timeout 300 python tools/kbench.py gemm --grid 64x64x128 --frames 8192 --svs 2562 --iters 20 2>/dev/null | tail -1 | cut -c1-200 >> $O/alone.txt
done
cat $O/pytest_svm.txt; cat $O/alone.txt; for f in $O/wal_*.json; do python tools/exp/show_bench.py $f $(basename $f .json); python -c "
import json,sys; d=json.load(open('$f')); print('   parity', d['parity']['label_calib_mismatch'], d['parity']['dec_ovo_max_abs_err'], 'e2e', d['hbm_frac_end_to_end'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], 'gemm chunk', d['gemm_roofline']['avg_chunk_ms'], 'crc', d['labels_crc32'])"; done; tail -3 $O/wal.err
