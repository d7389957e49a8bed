#!/bin/bash
# round-4 GPU session BG: uint8 kernel templated on read-compare-write, rule = 64x64x128 uint8 + derive -- tests, then this library against commit c3ef2bc's on one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4bg; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_projection_gpu.py tests/test_svm_gpu.py -x -q 2>&1 | tail -n 2
B="python bench.py --steps 8 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 1024"
for rep in 1 2; do for k in old new; do
  unset RML_LIB
  if [ $k = old ]; then export RML_LIB=$PWD/radar-ml_amd/libradarml_hip_c3ef2bc.so; fi
  timeout 900 $B > $O/${k}_$rep.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/${k}_$rep.json $k | grep -v "gate\|slice_mode" | cut -c1-125
done; done
