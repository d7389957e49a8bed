#!/bin/bash
# round-3 GPU session X: ring-schedule 128x128 GEMM beside the projection, now that the GEMM chain is the period at the Walabot grid
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3x; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 1024"
for rep in 1 2; do for v in base ring128 ring128nb3; do
  R=""; N=""; [ $v != base ] && R=1; [ $v = ring128nb3 ] && N=3
  RML_GEMM_RING128=$R RML_NBUF=$N timeout 900 $B > $O/${v}_$rep.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/${v}_$rep.json $v
  python -c "
import json; d=json.load(open('$O/${v}_$rep.json')); w=d['walabot_grid']
print('   launch ms', d['roofline']['avg_launch_ms'], w['roofline']['avg_launch_ms'], 'gemm chunk ms', d['gemm_roofline']['avg_chunk_ms'], w['gemm_roofline']['avg_chunk_ms'], 'e2e', d['hbm_frac_end_to_end'], w['hbm_frac_end_to_end'], 'parity', d['parity']['label_calib_mismatch'], w['parity']['label_calib_mismatch'])"
done; done
