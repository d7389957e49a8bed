#!/bin/bash
# round-4 GPU session AR: the all-exact scan of k_tile_flags with independent wide loads (it took 65-75 us per chunk on the GEMM stream) -- tests, pipelines, timeline
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4ar; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests/test_svm_gpu.py tests/test_nn_gpu.py -x -q -k "volumes or full_size or pipeline or preprocess or off_grid or general or mixed" 2>&1 | tail -n 3
B="python bench.py --steps 8 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 1024"
for rep in 1 2; do timeout 900 $B > $O/b$rep.json 2>> $O/b.err; python tools/exp/show_bench.py $O/b$rep.json run$rep | grep -v gate | cut -c1-130; done
cd /tmp
rocprofv3 --kernel-trace -d $R/$O/prof_wal -o k -- python $R/bench.py --steps 4 --warmup 2 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --no-slice --parity 256 --grid 22x31x176 --frames 262144 --no-walabot > /dev/null 2> $R/$O/wal.err
cd $R
python tools/timeline.py $(find $O/prof_wal -name "*.db" | head -1) --match k_project_lin --rows 44 > $O/r04_timeline_walabot.txt 2>&1
rm -rf $O/prof_wal
sed -n 3,16p $O/r04_timeline_walabot.txt | cut -c1-120; tail -1 $O/r04_timeline_walabot.txt
