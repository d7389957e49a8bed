#!/bin/bash
# round-3 GPU session B: exact-GEMM issue-schedule A/B (two-stage burst / ring burst / ring interleaved / two-stage staggered)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3b; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/oracle:$PYTHONPATH
timeout 600 python tools/gemm_ab.py exact --grid 64x64x128 --svs 2562 --frames 16384,17664 --rounds 4 > $O/exact_64.jsonl 2> $O/exact_64.err
timeout 600 python tools/gemm_ab.py exact --grid 22x31x176 --svs 2281 --frames 14336,16384 --rounds 4 > $O/exact_wal.jsonl 2> $O/exact_wal.err
timeout 600 python -m pytest tests/test_svm_gpu.py -x -q -k "big or 256" 2>&1 | tail -5 > $O/pytest_big.txt
cat $O/*.jsonl; tail -n 3 $O/*.err $O/pytest_big.txt
