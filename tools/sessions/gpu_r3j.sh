#!/bin/bash
# round-3 GPU session J: CU partition (GEMM on g CUs of every XCD with the ring kernel, projection stand-alone on the rest)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3j; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 1024"
for g in 0 4 6 8 10 12; do
  G=""; [ $g != 0 ] && G=$g
  RML_GEMM_CUS=$G timeout 900 $B > $O/part_$g.json 2>> $O/part.err
  python tools/exp/show_bench.py $O/part_$g.json g$g
  python -c "
import json; d=json.load(open('$O/part_$g.json')); w=d['walabot_grid']
print('   parity', d['parity']['label_calib_mismatch'], w['parity']['label_calib_mismatch'], 'u8 same', d['uint8_ingest']['identical_to_f32_ingest'], w['uint8_ingest']['identical_to_f32_ingest'], 'e2e', d['hbm_frac_end_to_end'], w['hbm_frac_end_to_end'], 'launch ms', d['roofline']['avg_launch_ms'], w['roofline']['avg_launch_ms'], 'gemm chunk ms', d['gemm_roofline']['avg_chunk_ms'], w['gemm_roofline']['avg_chunk_ms'])"
done
tail -3 $O/part.err
