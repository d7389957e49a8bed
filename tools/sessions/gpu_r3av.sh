#!/bin/bash
# round-3 GPU session AV: uint8 rows of 11 chunks (Walabot) in the 16-lane geometry (VALU cross-lane steps) vs their own 11-lane geometry
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3av; mkdir -p $O
timeout 900 python -m pytest tests/test_projection_gpu.py -x -q 2>&1 | tail -n 3
for rep in 1 2 3; do for g in 1 0; do
  echo "== PADGEOM=$g"; RML_U8_PADGEOM=$g timeout 300 python tools/kbench.py proj --u8 --grid 22x31x176 --frames 32768 2>&1 | grep "codes+stats" | cut -c40-230
done; done
B="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 1024 --ingest u8 --grid 22x31x176 --frames 262144 --no-walabot"
for rep in 1 2; do for g in 1 0; do
  RML_U8_PADGEOM=$g timeout 900 $B > $O/g${g}_$rep.json 2>> $O/b.err
  python -c "
import json; d=json.load(open('$O/g${g}_$rep.json')); print('padgeom$g', round(d['value']/1e6,3), 'launch', d['roofline']['avg_launch_ms'], 'e2e', d['hbm_frac_end_to_end'], d['parity']['label_calib_mismatch'], d['labels_crc32'])"
done; done
