#!/bin/bash
# round-3 GPU session D: rotated K loop (deferred MFMA group behind the barrier) vs the unrotated ring kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3d; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/oracle:$PYTHONPATH
timeout 900 python -m pytest tests/test_svm_gpu.py -x -q 2>&1 | tail -8 > $O/pytest_svm.txt
for v in rot norot rot norot; do
  L=""; [ $v = norot ] && L=$PWD/radar-ml_amd/libradarml_hip_norot.so
  RML_LIB=$L timeout 600 python tools/gemm_ab.py exact --grid 64x64x128 --svs 2562 --frames 17664,23808 --rounds 3 >> $O/exact_64_$v.jsonl 2>> $O/exact_64.err
  RML_LIB=$L timeout 600 python tools/gemm_ab.py exact --grid 22x31x176 --svs 2281 --frames 14336,21760 --rounds 3 >> $O/exact_wal_$v.jsonl 2>> $O/exact_wal.err
done
RML_LIB= timeout 900 python tools/gemm_ab.py digits --grid 64x64x128 --svs 2562 --frames 17664 --rounds 2 > $O/digits_64_rot.jsonl 2> $O/digits_64.err
RML_LIB=$PWD/radar-ml_amd/libradarml_hip_norot.so timeout 900 python tools/gemm_ab.py digits --grid 64x64x128 --svs 2562 --frames 17664 --rounds 2 > $O/digits_64_norot.jsonl 2>> $O/digits_64.err
cat $O/pytest_svm.txt; for f in $O/*.jsonl; do echo "== $f"; python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    if d["what"]=="exact": print(d["grid"],d["N"],"ring",d["ring"]["ms"],d["ring"]["frac_of_3944"],"2stage",d["big2stage"]["ms"],"bits",d["ring_equals_2stage_bits"])
    else: print(d["grid"],d["N"],"digits",d["digits"],"f64",d["f64"]["ms"],"diff",d["max_abs_dec_diff"])
PY
done; tail -n 2 $O/*.err
