#!/bin/bash
# round-4 GPU session X: k_project_lin ablations -- rolling refill x per-plane stores removed (timing only), Walabot grid, codes + statistics
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in "" _ns _roll _rollns; do
    printf "%-8s " "lib$v"
    RML_LIB=$PWD/radar-ml_amd/libradarml_hip$v.so timeout 300 python tools/kbench.py proj --grid 22x31x176 --frames 16384 | grep "codes+stats only" | cut -c60-200
  done
done
