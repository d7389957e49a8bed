#!/bin/bash
# round-4 GPU session AD: code rows with non-temporal stores (variant build RML_CODE_NT=1), alone and in the pipeline
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4ad; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in "" _nt; do
    L=$PWD/radar-ml_amd/libradarml_hip$v.so
    printf "%-8s walabot f32   " "lib$v"; RML_LIB=$L timeout 300 python tools/kbench.py proj --grid 22x31x176 --frames 16384 | grep "codes+stats only" | cut -c75-200
    printf "%-8s 64x64x128 f32 (pipeline config) " "lib$v"; RML_WAVE_SHARE=1 RML_LIB=$L timeout 300 python tools/kbench.py proj --grid 64x64x128 --frames 8192 | grep "codes+stats only" | cut -c75-200
  done
done
B="python bench.py --steps 8 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --no-slice --parity 1024"
for rep in 1 2; do for v in "" _nt; do
  RML_LIB=$PWD/radar-ml_amd/libradarml_hip$v.so timeout 900 $B > $O/b${v}_$rep.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/b${v}_$rep.json "lib$v" | cut -c1-150
done; done
