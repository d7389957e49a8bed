#!/bin/bash
# round-3 GPU session N: second projection stream for odd chunks; digit path with both parked levels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3n; mkdir -p $O
timeout 900 python -m pytest tests/test_svm_gpu.py tests/test_capi_gpu.py -x -q -s 2>&1 | grep "digits\|passed\|failed\|Error" | tail -20 > $O/pytest_svm.txt
timeout 900 python tools/gemm_ab.py digits --grid 64x64x128 --svs 2562 --frames 17664 --rounds 2 > $O/digits_64.jsonl 2> $O/digits.err
B="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 1024"
for rep in 1 2; do
  for ps in 1 2; do
    P=""; [ $ps = 2 ] && P=2
    RML_PROJ_STREAMS=$P timeout 900 $B > $O/ps${ps}_$rep.json 2>> $O/b.err
    python tools/exp/show_bench.py $O/ps${ps}_$rep.json ps$ps
    python -c "
import json; d=json.load(open('$O/ps${ps}_$rep.json')); w=d['walabot_grid']
print('   parity', d['parity']['label_calib_mismatch'], w['parity']['label_calib_mismatch'], 'e2e', d['hbm_frac_end_to_end'], w['hbm_frac_end_to_end'], 'launch ms', d['roofline']['avg_launch_ms'], w['roofline']['avg_launch_ms'], 'gemm chunk ms', d['gemm_roofline']['avg_chunk_ms'], w['gemm_roofline']['avg_chunk_ms'], 'crc', d['labels_crc32'], w['labels_crc32'])"
  done
done
cat $O/pytest_svm.txt; cut -c1-400 $O/digits_64.jsonl
