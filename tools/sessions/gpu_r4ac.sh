#!/bin/bash
# round-4 GPU session AC: CNN front door with its fallback launches on the second stream; k_set_int merged into k_tile_flags; chain + pipeline numbers
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4ac; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_nn_gpu.py tests/test_svm_gpu.py tests/test_capi_gpu.py -x -q 2>&1 | tail -n 5
for rep in 1 2; do
  timeout 300 python tools/dnn_chain.py
  timeout 300 python tools/dnn_chain.py --u8
done
B="python bench.py --steps 8 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 1024"
timeout 900 $B > $O/bench.json 2>> $O/b.err
python tools/exp/show_bench.py $O/bench.json now | cut -c1-170
