#!/bin/bash
# round-3 GPU session A: digit-path parity, exact-GEMM A/B (exp, ring), digits vs f64 rate
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3a; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/oracle:$PYTHONPATH
timeout 900 python -m pytest tests/test_svm_gpu.py -x -q -s 2>&1 | tail -40 > $O/pytest_svm.txt
timeout 600 python tools/gemm_ab.py exact --grid 64x64x128 --svs 2562 --frames 16384,17664 --rounds 4 > $O/exact_64.jsonl 2> $O/exact_64.err
timeout 600 python tools/gemm_ab.py exact --grid 22x31x176 --svs 2281 --frames 16384,14336,21760 --rounds 4 > $O/exact_wal.jsonl 2> $O/exact_wal.err
RML_LIB=$PWD/radar-ml_amd/libradarml_hip_libm.so timeout 600 python tools/gemm_ab.py exact --grid 64x64x128 --svs 2562 --frames 16384 --rounds 4 > $O/exact_64_libm.jsonl 2> $O/exact_64_libm.err
timeout 900 python tools/gemm_ab.py digits --grid 64x64x128 --svs 2562 --frames 16384 --rounds 2 > $O/digits_64.jsonl 2> $O/digits_64.err
timeout 900 python tools/gemm_ab.py digits --grid 22x31x176 --svs 2281 --frames 14336 --rounds 2 > $O/digits_wal.jsonl 2> $O/digits_wal.err
tail -5 $O/pytest_svm.txt; cat $O/*.jsonl; tail -3 $O/*.err
