#!/bin/bash
# round-3 GPU session M: tile decision + predicated second pass on the GEMM stream; optional third workspace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3m; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest_gpu.txt
B="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 1024"
for rep in 1 2; do
  for nb in 2 3; do
    RML_NBUF=$nb timeout 900 $B > $O/nbuf${nb}_$rep.json 2>> $O/b.err
    python tools/exp/show_bench.py $O/nbuf${nb}_$rep.json nbuf$nb
    python -c "
import json; d=json.load(open('$O/nbuf${nb}_$rep.json')); w=d['walabot_grid']
print('   parity', d['parity']['label_calib_mismatch'], w['parity']['label_calib_mismatch'], 'u8 same', d['uint8_ingest']['identical_to_f32_ingest'], w['uint8_ingest']['identical_to_f32_ingest'], 'e2e', d['hbm_frac_end_to_end'], w['hbm_frac_end_to_end'], 'launch ms', d['roofline']['avg_launch_ms'], w['roofline']['avg_launch_ms'], 'gemm chunk ms', d['gemm_roofline']['avg_chunk_ms'], w['gemm_roofline']['avg_chunk_ms'])"
  done
done
cat $O/pytest_gpu.txt
