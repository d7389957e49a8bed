#!/bin/bash
# round-4 GPU session AB: what do the code-row writes cost the projection kernels?  variant builds RML_EMIT_ABL: 1 = no code stores, 2 = all frames'
# code rows on 64 rows (L2-resident: the store instructions stay, the HBM writes go); alone, codes + statistics
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
  for v in "" _emit1 _emit2; do
    L=$PWD/radar-ml_amd/libradarml_hip$v.so
    printf "%-10s walabot f32   " "lib$v"; RML_LIB=$L timeout 300 python tools/kbench.py proj --grid 22x31x176 --frames 16384 | grep "codes+stats only" | cut -c75-200
    printf "%-10s walabot u8    " "lib$v"; RML_LIB=$L timeout 300 python tools/kbench.py proj --grid 22x31x176 --frames 32768 --u8 | grep "codes+stats only" | cut -c75-200
    printf "%-10s 64x64x128 u8  " "lib$v"; RML_LIB=$L timeout 300 python tools/kbench.py proj --grid 64x64x128 --frames 8192 --u8 | grep "codes+stats only" | cut -c75-200
    printf "%-10s 64x64x128 f32 (pipeline config) " "lib$v"; RML_WAVE_SHARE=1 RML_LIB=$L timeout 300 python tools/kbench.py proj --grid 64x64x128 --frames 8192 | grep "codes+stats only" | cut -c75-200
  done
done
