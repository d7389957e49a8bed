#!/bin/bash
# round-3 GPU session AB: three streams + three workspaces on both grids; chunk sizes; timeline
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3ab; mkdir -p $O
export TMPDIR=/tmp
BW="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 1024 --grid 22x31x176 --frames 262144 --no-walabot"
BH="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 1024 --no-walabot"
run() {  # tag cmdvar split nbuf chunk
  local cmd="$BW"; [ $2 = H ] && cmd="$BH"
  RML_PIPE_SPLIT=$3 RML_NBUF=$4 RML_CHUNK=$5 timeout 600 $cmd > $O/$1.json 2>> $O/b.err
  python -c "
import json; d=json.load(open('$O/$1.json'))
print('$1', 'M frames/s', round(d['value']/1e6,3), 'launch ms', d['roofline']['avg_launch_ms'], 'gemm chunk ms', d['gemm_roofline']['avg_chunk_ms'], 'e2e', d['hbm_frac_end_to_end'], 'kernel', d['roofline']['frac'], 'parity', d['parity']['label_calib_mismatch'], d['labels_crc32'])"
}
for rep in 1 2; do
  run H_base_$rep H 0 "" ""
  run H_split_nb3_$rep H 1 3 ""
  run H_nosplit_nb3_$rep H 0 3 ""
  run W_base_$rep W 0 "" ""
  run W_split_nb3_$rep W 1 3 ""
  run W_split_nb3_c16k_$rep W 1 3 16384
  run W_split_nb3_c6k_$rep W 1 3 6144
done
R=$PWD
cd /tmp
RML_PIPE_SPLIT=1 RML_NBUF=3 timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof_wal -o k -- python $R/bench.py --steps 4 --warmup 2 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 256 --grid 22x31x176 --frames 262144 --no-walabot > $R/$O/wal_prof.json 2> $R/$O/wal_prof.err
cd $R
python tools/timeline.py $O/prof_wal/k_results.db --match k_project_lin --rows 40 > $O/timeline.txt 2>&1
rm -rf $O/prof_wal
cat $O/timeline.txt | cut -c1-120
