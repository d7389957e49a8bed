#!/bin/bash
# round-3 GPU session E: digits with the parked top group (no register remainders), rotated loop
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3e; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/oracle:$PYTHONPATH
timeout 900 python -m pytest tests/test_svm_gpu.py -x -q 2>&1 | tail -8 > $O/pytest_svm.txt
timeout 900 python tools/gemm_ab.py digits --grid 64x64x128 --svs 2562 --frames 17664,23808 --rounds 2 > $O/digits_64.jsonl 2> $O/digits_64.err
timeout 900 python tools/gemm_ab.py digits --grid 22x31x176 --svs 2281 --frames 14336,21760 --rounds 2 > $O/digits_wal.jsonl 2> $O/digits_wal.err
timeout 600 python tools/gemm_ab.py exact --grid 64x64x128 --svs 2562 --frames 23808 --rounds 3 > $O/exact_64.jsonl 2> $O/exact_64.err
cat $O/pytest_svm.txt; for f in $O/*.jsonl; do echo "== $f"; python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    if d["what"]=="exact": print(d["grid"],d["N"],"ring",d["ring"]["ms"],d["ring"]["frac_of_3944"],"2stage",d["big2stage"]["ms"],"128",d["tile128"]["ms"],"bits",d["ring_equals_2stage_bits"],"d128",d["max_abs_diff_128_vs_256"])
    else: print(d["grid"],d["N"],"digits",d["digits"],"POPs",d["digits_int8_POPs"],"f64",d["f64"]["ms"],"diff",d["max_abs_dec_diff"])
PY
done; tail -n 2 $O/*.err
