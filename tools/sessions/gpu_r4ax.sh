#!/bin/bash
# round-4 GPU session AX: read-compare-write of the code rows as the default (rml_code_rmw: rows <= 1/16 of the frame) -- new test, CNN tests, A/B against RML_CODE_RMW=0
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4ax; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_svm_gpu.py -x -q -k "read_compare_write or fused or full_size" 2>&1 | tail -n 3
timeout 900 python -m pytest tests/test_nn_gpu.py tests/test_projection_gpu.py -x -q 2>&1 | tail -n 3
B="python bench.py --steps 8 --warmup 3 --no-general --no-sgan --no-cpu --no-pmc --parity 1024"
for rep in 1 2; do for k in 0 d; do
  if [ $k = 0 ]; then export RML_CODE_RMW=0; else unset RML_CODE_RMW; fi
  timeout 900 $B > $O/rmw${k}_$rep.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/rmw${k}_$rep.json rmw$k | grep -v "gate\|slice_mode" | cut -c1-125
done; done
