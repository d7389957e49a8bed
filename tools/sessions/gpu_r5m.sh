#!/bin/bash
# round-5 GPU session M: kernel timelines of the fused pipeline (Walabot grid, 64x64x128) with the closing library, and the wall time of the default bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5m; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace -d $R/$O/prof_wal -o k -- python $R/bench.py --steps 4 --warmup 2 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --no-slice --parity 256 --grid 22x31x176 --frames 262144 --no-walabot > /dev/null 2> $R/$O/wal.err
rocprofv3 --kernel-trace -d $R/$O/prof_head -o k -- python $R/bench.py --steps 4 --warmup 2 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --no-slice --parity 256 --no-walabot > /dev/null 2> $R/$O/head.err
cd $R
python tools/timeline.py $(find $O/prof_wal -name "*.db" | head -1) --match k_project_lin --rows 44 > $O/r05_timeline_walabot.txt 2>&1
python tools/timeline.py $(find $O/prof_head -name "*.db" | head -1) --match k_project_wave --rows 30 > $O/r05_timeline_headline.txt 2>&1
rm -rf $O/prof_wal $O/prof_head
sed -n 3,14p $O/r05_timeline_walabot.txt | cut -c1-120; tail -1 $O/r05_timeline_walabot.txt; tail -1 $O/r05_timeline_headline.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err ) 2>&1 | tail -3
tail -c 600 $O/bench.log
python -m pytest tests/test_nn_gpu.py tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -2
