#!/bin/bash
# round-3 GPU session AN: k_svm_gemm_lite (40 KB exact 128x128 GEMM) -- bit equality, alone, and beside the projection (2 and 3 streams)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3an; mkdir -p $O
timeout 900 python -m pytest tests/test_svm_gpu.py tests/test_capi_gpu.py -x -q 2>&1 | tail -n 4
BW="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 1024 --grid 22x31x176 --frames 262144 --no-walabot"
BH="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 1024 --no-walabot"
run() {
  local cmd="$BW"; [ $2 = H ] && cmd="$BH"
  RML_GEMM_LITE=$3 RML_PIPE_SPLIT=$4 RML_NBUF=$5 timeout 600 $cmd > $O/$1.json 2>> $O/b.err
  python -c "
import json; d=json.load(open('$O/$1.json'))
print('$1', round(d['value']/1e6,3), 'launch', d['roofline']['avg_launch_ms'], 'gemm', d['gemm_roofline']['avg_chunk_ms'], 'alone', d['gemm_roofline']['alone']['ms'], 'e2e', d['hbm_frac_end_to_end'], d['parity']['label_calib_mismatch'], d['labels_crc32'])"
}
for rep in 1 2 3; do
  run W_base_$rep W "" "" ""
  run W_lite_$rep W 1 "" ""
  run W_lite_split3_$rep W 1 1 3
  run H_base_$rep H "" "" ""
  run H_lite_$rep H 1 "" ""
done
