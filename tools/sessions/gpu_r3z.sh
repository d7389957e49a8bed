#!/bin/bash
# round-3 GPU session Z: three-stream pipeline (small kernels off the GEMM stream) vs two streams; timeline at the Walabot grid
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3z; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_svm_gpu.py tests/test_capi_gpu.py tests/test_dist_gpu.py -x -q 2>&1 | tail -n 4
B="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 2048"
for rep in 1 2 3; do for v in 1 0; do
  RML_PIPE_SPLIT=$v timeout 900 $B > $O/split${v}_$rep.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/split${v}_$rep.json split$v
  python -c "
import json; d=json.load(open('$O/split${v}_$rep.json')); w=d['walabot_grid']
print('   launch ms', d['roofline']['avg_launch_ms'], w['roofline']['avg_launch_ms'], 'gemm chunk ms', d['gemm_roofline']['avg_chunk_ms'], w['gemm_roofline']['avg_chunk_ms'], 'e2e', d['hbm_frac_end_to_end'], w['hbm_frac_end_to_end'], 'parity', d['parity']['label_calib_mismatch'], w['parity']['label_calib_mismatch'], d['labels_crc32'], w['labels_crc32'])"
done; done
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof_wal -o k -- python $R/bench.py --steps 4 --warmup 2 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 256 --grid 22x31x176 --frames 262144 --no-walabot > $R/$O/wal_prof.json 2> $R/$O/wal_prof.err
cd $R
python tools/timeline.py $O/prof_wal/k_results.db --match k_project_lin --rows 44 > $O/timeline_walabot_split.txt 2>&1
rm -rf $O/prof_wal
cat $O/timeline_walabot_split.txt | cut -c1-130
