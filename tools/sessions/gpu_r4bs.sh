#!/bin/bash
# round-4 GPU session BS: derive -> slice with the old code words loaded alongside the gather's batches -- tests, both grids with RML_CODE_RMW unset / 0 / 1, three interleaved rounds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4bs; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_svm_gpu.py tests/test_projection_gpu.py -x -q -k "slice or derive or read_compare" 2>&1 | tail -n 2
W="python bench.py --steps 8 --warmup 3 --grid 22x31x176 --frames 262144 --no-walabot --no-u8 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 256"
H="python bench.py --steps 8 --warmup 3 --no-walabot --no-u8 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 256"
for rep in 1 2 3; do for k in d 0 1; do
  if [ $k = d ]; then unset RML_CODE_RMW; else export RML_CODE_RMW=$k; fi
  timeout 900 $H > $O/h${k}_$rep.json 2>> $O/b.err
  echo -n "64x64x128 rmw=$k: "; python tools/exp/show_bench.py $O/h${k}_$rep.json x | grep "derive" | cut -c1-100
  timeout 900 $W > $O/w${k}_$rep.json 2>> $O/b.err
  echo -n "walabot   rmw=$k: "; python tools/exp/show_bench.py $O/w${k}_$rep.json x | grep "derive" | cut -c1-100
done; done
