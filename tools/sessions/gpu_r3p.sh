#!/bin/bash
# round-3 GPU session P: wave issue priority (s_setprio) of the GEMM / projection waves in the fused pipeline; kernel timeline of the
# Walabot pipeline (where do the 0.12 ms per chunk beside the projection launches go)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3p; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 1024"
timeout 600 python -m pytest tests/test_projection_gpu.py -x -q -k "nan" 2>&1 | tail -n 3
for rep in 1 2; do
  for v in base pipedig priog priop; do
    L=""; [ $v = priog -o $v = priop ] && L=$PWD/radar-ml_amd/libradarml_hip_$v.so
    PD=""; [ $v = pipedig ] && PD=1
    RML_PIPE_DIGITS=$PD RML_LIB=$L timeout 900 $B > $O/${v}_$rep.json 2>> $O/b.err
    python tools/exp/show_bench.py $O/${v}_$rep.json $v
  done
done
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof_wal -o k -- python $R/bench.py --steps 4 --warmup 2 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 256 --grid 22x31x176 --frames 262144 --no-walabot > $R/$O/wal_prof.json 2> $R/$O/wal_prof.err
cd $R
python tools/timeline.py $O/prof_wal/k_results.db --match k_project_lin --rows 60 > $O/timeline_walabot.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof_hl -o k -- python $R/bench.py --steps 4 --warmup 2 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 256 --no-walabot > $R/$O/hl_prof.json 2> $R/$O/hl_prof.err
cd $R
python tools/timeline.py $O/prof_hl/k_results.db --match k_project_wave --rows 40 > $O/timeline_headline.txt 2>&1
rm -rf $O/prof_wal $O/prof_hl
cat $O/timeline_walabot.txt | cut -c1-150; tail -3 $O/timeline_headline.txt
