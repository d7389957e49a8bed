#!/bin/bash
# round-3 GPU session AT: k_svm_gemm_tall (256 SVs x 128 samples, 72 KB) -- bit equality, alone, and beside the projection
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3at; mkdir -p $O
timeout 900 python -m pytest tests/test_svm_gpu.py tests/test_capi_gpu.py -x -q 2>&1 | tail -n 4
for t in "" 2; do echo "== RML_GEMM_TALL=$t (RML_GEMM_BIG=0)"; RML_GEMM_BIG=0 RML_GEMM_TALL=$t timeout 300 python tools/kbench.py gemm --grid 22x31x176 --frames 8192 --svs 2000 --iters 10 2>&1 | grep -v amdgpu | cut -c1-250; RML_GEMM_BIG=0 RML_GEMM_TALL=$t timeout 300 python tools/kbench.py gemm --grid 64x64x128 --frames 8192 --svs 2562 --iters 10 2>&1 | grep -v amdgpu | cut -c1-250; done
BW="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 1024 --grid 22x31x176 --frames 262144 --no-walabot"
BH="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 1024 --no-walabot"
run() {
  local cmd="$BW"; [ $2 = H ] && cmd="$BH"
  RML_GEMM_TALL=$3 timeout 600 $cmd > $O/$1.json 2>> $O/b.err
  python -c "
import json; d=json.load(open('$O/$1.json'))
print('$1', round(d['value']/1e6,3), 'launch', d['roofline']['avg_launch_ms'], 'gemm', d['gemm_roofline']['avg_chunk_ms'], 'e2e', d['hbm_frac_end_to_end'], 'kernel', d['roofline']['frac'], d['parity']['label_calib_mismatch'], d['labels_crc32'])"
}
for rep in 1 2 3; do
  run W_base_$rep W ""
  run W_tall_$rep W 1
  run H_base_$rep H ""
  run H_tall_$rep H 1
done
