#!/bin/bash
# round-4 GPU session AF: k_pre3 ablations (timing only): 1 = no horizontal pass, 2 = no vertical arithmetic, 3 = no output stores
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do for v in "" _p1 _p2 _p3; do printf "%-7s " "lib$v"; RML_LIB=$PWD/radar-ml_amd/libradarml_hip$v.so timeout 300 python tools/pre3_bench.py; done; done
