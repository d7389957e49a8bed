#!/bin/bash
# round-3 GPU session C: unified ring kernel (exact + digits): parity, A/B, balanced XCD map
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3c; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/oracle:$PYTHONPATH
timeout 900 python -m pytest tests/test_svm_gpu.py -x -q 2>&1 | tail -8 > $O/pytest_svm.txt
timeout 600 python tools/gemm_ab.py exact --grid 64x64x128 --svs 2562 --frames 16384,17408,17664,23808 --rounds 3 > $O/exact_64.jsonl 2> $O/exact_64.err
timeout 600 python tools/gemm_ab.py exact --grid 22x31x176 --svs 2281 --frames 14336,16384,21760 --rounds 3 > $O/exact_wal.jsonl 2> $O/exact_wal.err
timeout 900 python tools/gemm_ab.py digits --grid 64x64x128 --svs 2562 --frames 16384 --rounds 2 > $O/digits_64.jsonl 2> $O/digits_64.err
timeout 900 python tools/gemm_ab.py digits --grid 22x31x176 --svs 2281 --frames 14336 --rounds 2 > $O/digits_wal.jsonl 2> $O/digits_wal.err
cat $O/pytest_svm.txt $O/*.jsonl; tail -n 3 $O/*.err
