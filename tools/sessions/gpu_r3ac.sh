#!/bin/bash
# round-3 GPU session AC: three streams + three workspaces vs the two-stream default, five interleaved rounds per grid
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3ac; mkdir -p $O
BW="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 512 --grid 22x31x176 --frames 262144 --no-walabot"
BH="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 512 --no-walabot"
run() {
  local cmd="$BW"; [ $2 = H ] && cmd="$BH"
  RML_PIPE_SPLIT=$3 RML_NBUF=$4 timeout 600 $cmd > $O/$1.json 2>> $O/b.err
  python -c "
import json; d=json.load(open('$O/$1.json'))
print('$1', round(d['value']/1e6,3), 'launch', d['roofline']['avg_launch_ms'], 'gemm', d['gemm_roofline']['avg_chunk_ms'], 'e2e', d['hbm_frac_end_to_end'], d['parity']['label_calib_mismatch'], d['labels_crc32'])"
}
for rep in 1 2 3 4 5; do
  run W_base_$rep W 0 ""
  run W_split3_$rep W 1 3
  run H_base_$rep H 0 ""
  run H_split3_$rep H 1 3
done
