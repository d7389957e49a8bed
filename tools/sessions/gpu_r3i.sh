#!/bin/bash
# round-3 GPU session I: the linear-plane projection kernel (k_project_lin): parity, stand-alone rate, pipeline A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3i; mkdir -p $O
timeout 1500 python -m pytest tests/test_projection_gpu.py -x -q 2>&1 | tail -6 > $O/pytest_proj.txt
for n in 8192 16384; do
  for l in 1 0 1 0; do
    echo "LINPLANE=$l frames=$n" >> $O/proj.txt
    RML_LINPLANE=$l timeout 300 python tools/kbench.py proj --grid 22x31x176 --frames $n --iters 20 2>/dev/null | cut -c1-230 >> $O/proj.txt
    echo "LINPLANE=$l frames=$n share" >> $O/proj.txt
    RML_WAVE_SHARE=1 RML_LINPLANE=$l timeout 300 python tools/kbench.py proj --grid 22x31x176 --frames $n --iters 20 2>/dev/null | tail -1 | cut -c1-230 >> $O/proj.txt
  done
done
B="python bench.py --grid 22x31x176 --frames 262144 --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 1024"
for rep in 1 2; do
  for l in 1 0; do
    RML_LINPLANE=$l timeout 900 $B > $O/wal_lin${l}_$rep.json 2>> $O/wal.err
  done
done
cat $O/pytest_proj.txt; cat $O/proj.txt; for f in $O/wal_*.json; do python tools/exp/show_bench.py $f $(basename $f .json); python -c "
import json,sys; d=json.load(open('$f')); print('   parity', d['parity']['label_calib_mismatch'], d['parity']['dec_ovo_max_abs_err'], 'e2e', d['hbm_frac_end_to_end'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])"; done; tail -3 $O/wal.err
