#!/bin/bash
# round-4 GPU session AQ: kernel timelines of the two max pipelines with the code stage in place (steady state of a step)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4aq; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
B="python $R/bench.py --steps 4 --warmup 2 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --no-slice --parity 256"
rocprofv3 --kernel-trace -d $R/$O/prof_wal -o k -- $B --grid 22x31x176 --frames 262144 --no-walabot > /dev/null 2> $R/$O/wal.err
rocprofv3 --kernel-trace -d $R/$O/prof_hl -o k -- $B --no-walabot > /dev/null 2> $R/$O/hl.err
cd $R
python tools/timeline.py $(find $O/prof_wal -name "*.db" | head -1) --match k_project_lin --rows 44 > $O/r04_timeline_walabot.txt 2>&1
python tools/timeline.py $(find $O/prof_hl -name "*.db" | head -1) --match k_project_wave --rows 40 > $O/r04_timeline_headline.txt 2>&1
rm -rf $O/prof_wal $O/prof_hl
head -50 $O/r04_timeline_walabot.txt | cut -c1-150
