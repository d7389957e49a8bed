#!/bin/bash
# round-3 GPU session S: the new uint8 kernel alone; uint8 pipeline with the 128x128 GEMM beside the projection vs the ring GEMM in whole rounds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3s; mkdir -p $O
for x in 1 0; do
  echo "== RML_U8_XLANE=$x"
  RML_U8_XLANE=$x timeout 300 python tools/kbench.py proj --u8 --grid 64x64x128 --frames 8192 2>&1 | grep -v amdgpu.ids | cut -c1-260
  RML_U8_XLANE=$x timeout 300 python tools/kbench.py proj --u8 --grid 64x64x256 --frames 4096 2>&1 | grep -v amdgpu.ids | cut -c1-260
done
B="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 1024 --ingest u8"
for rep in 1 2; do for v in base pipegemm prio; do
  PG=""; [ $v = pipegemm ] && PG=1
  L=""; [ $v = prio ] && L=$PWD/radar-ml_amd/libradarml_hip_priog.so
  RML_LIB=$L RML_PIPE_GEMM=$PG timeout 900 $B > $O/${v}_$rep.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/${v}_$rep.json u8-$v
  python -c "
import json; d=json.load(open('$O/${v}_$rep.json')); w=d['walabot_grid']
print('   launch ms', d['roofline']['avg_launch_ms'], w['roofline']['avg_launch_ms'], 'gemm chunk ms', d['gemm_roofline']['avg_chunk_ms'], w['gemm_roofline']['avg_chunk_ms'], 'e2e', d['hbm_frac_end_to_end'], w['hbm_frac_end_to_end'])"
done; done
