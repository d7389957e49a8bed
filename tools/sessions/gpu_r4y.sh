#!/bin/bash
# round-4 GPU session Y: k_project_lin ablations -- which per-CU resource bounds it (timing only): 1 = no xz LDS image / fold, 2 = no xy strip, 3 = both
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
  for v in "" _abl1 _abl2 _abl3; do
    for pc in 2 1; do
      printf "%-8s percu=%s " "lib$v" $pc
      RML_WAVE_PERCU=$pc RML_LIB=$PWD/radar-ml_amd/libradarml_hip$v.so timeout 300 python tools/kbench.py proj --grid 22x31x176 --frames 16384 | grep "codes+stats only" | cut -c75-200
    done
  done
done
