#!/bin/bash
# round-4 GPU session AT: k_fc1_splitk taking the LAST sample tiles first (the features the trunk wrote last may still be in the Infinity Cache), variant build, in the chain
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 300 env RML_LIB=$PWD/radar-ml_amd/libradarml_hip_rev.so python -m pytest tests/test_nn_gpu.py -x -q -k "dense or kblock" 2>&1 | tail -n 2
for rep in 1 2 3; do for v in "" _rev; do printf "%-8s " "lib$v"; RML_LIB=$PWD/radar-ml_amd/libradarml_hip$v.so timeout 300 python tools/dnn_chain.py; done; done
R=$PWD; O=gpurun_out/r4at; mkdir -p $O
cd /tmp && RML_LIB=$R/radar-ml_amd/libradarml_hip_rev.so timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o dnn -- python $R/tools/dnn_chain.py --steps 3 > $R/$O/prof.log 2>&1; cd $R
python tools/prof_summary.py stats $O/prof/dnn_results.db | grep "fc1\|trunk" | cut -c1-140; rm -rf $O/prof
