#!/bin/bash
# round-4 GPU session F: derive -> slice -> SVM pipeline, pairing (8-wave k_derive_slice workgroup per CU beside 128x128 GEMM
# workgroups, 8192-frame chunks) against taking turns with the ring GEMM
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_projection_gpu.py tests/test_svm_gpu.py -x -q -k "derive or slice" 2>&1 | tail -n 3
B="python bench.py --steps 6 --warmup 2 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 1024"
for rep in 1 2; do for v in 1 0; do
  RML_DERIVE_PIPE=$v timeout 900 $B > $O/pipe${v}_$rep.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/pipe${v}_$rep.json pipe$v | grep -v slice_mode
done; done
for ch in 4096 16384; do
  RML_CHUNK=$ch timeout 900 $B > $O/chunk$ch.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/chunk$ch.json chunk$ch | grep derive
done
