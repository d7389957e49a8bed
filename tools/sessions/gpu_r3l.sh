#!/bin/bash
# round-3 GPU session L: chunk size of the Walabot pipeline with the linear-plane kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3l; mkdir -p $O
B="python bench.py --grid 22x31x176 --frames 262144 --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 1024"
for rep in 1 2; do
  for c in 8192 12288 16384 6144; do
    RML_CHUNK=$c timeout 900 $B > $O/wal_c${c}_$rep.json 2>> $O/wal.err
    python tools/exp/show_bench.py $O/wal_c${c}_$rep.json c$c
    python -c "
import json; d=json.load(open('$O/wal_c${c}_$rep.json')); print('   e2e', d['hbm_frac_end_to_end'], 'launch', d['roofline']['avg_launch_ms'], d['roofline']['frames_per_launch'], 'gemm chunk', d['gemm_roofline']['avg_chunk_ms'], 'ms/step', d['ms_per_step'])"
  done
done
