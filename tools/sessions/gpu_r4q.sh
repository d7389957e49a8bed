#!/bin/bash
# round-4 GPU session Q: tapered chunks (quarter-size first and last chunk) in the paired pipelines, A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4q; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_svm_gpu.py tests/test_capi_gpu.py -x -q 2>&1 | tail -n 4
B="python bench.py --steps 8 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 1024"
for rep in 1 2; do for v in 1 0; do
  RML_TAPER=$v timeout 900 $B > $O/taper${v}_$rep.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/taper${v}_$rep.json taper$v | grep -v slice_mode | cut -c1-140
done; done
for ch in 16384; do
  RML_CHUNK=$ch timeout 900 $B > $O/chunk$ch.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/chunk$ch.json chunk$ch | grep -v slice_mode | cut -c1-140
done
