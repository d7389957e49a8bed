#!/bin/bash
# round-3 GPU session AA: three-stream pipeline with three workspaces and one GEMM workgroup per CU (LDS pad), Walabot grid
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3aa; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 1024 --grid 22x31x176 --frames 262144 --no-walabot"
run() {  # tag split nbuf ldsx
  RML_PIPE_SPLIT=$2 RML_NBUF=$3 RML_GEMM_LDS_EXTRA=$4 timeout 600 $B > $O/$1.json 2>> $O/b.err
  python -c "
import json; d=json.load(open('$O/$1.json'))
print('$1', 'M frames/s', round(d['value']/1e6,3), 'launch ms', d['roofline']['avg_launch_ms'], 'gemm chunk ms', d['gemm_roofline']['avg_chunk_ms'], 'e2e', d['hbm_frac_end_to_end'], 'parity', d['parity']['label_calib_mismatch'], d['labels_crc32'])"
}
for rep in 1 2; do
  run base_$rep 0 "" ""
  run split_nb3_$rep 1 3 ""
  run split_nb3_lds_$rep 1 3 12288
  run split_nb2_lds_$rep 1 "" 12288
  run nosplit_nb3_lds_$rep 0 3 12288
  run nosplit_lds_$rep 0 "" 12288
done
R=$PWD
cd /tmp
RML_PIPE_SPLIT=1 RML_NBUF=3 RML_GEMM_LDS_EXTRA=12288 timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof_wal -o k -- python $R/bench.py --steps 4 --warmup 2 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 256 --grid 22x31x176 --frames 262144 --no-walabot > $R/$O/wal_prof.json 2> $R/$O/wal_prof.err
cd $R
python tools/timeline.py $O/prof_wal/k_results.db --match k_project_lin --rows 40 > $O/timeline.txt 2>&1
rm -rf $O/prof_wal
cat $O/timeline.txt | cut -c1-120
