#!/bin/bash
# round-4 GPU session BK: conv1 tests; chunk size of the derive -> slice -> SVM pairing at both grids (RML_CHUNK), six interleaved rounds (each run is its own process: placement varies per process)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4bk; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_nn_gpu.py -x -q -k "conv1" 2>&1 | tail -n 2
W="python bench.py --steps 8 --warmup 3 --grid 22x31x176 --frames 262144 --no-walabot --no-u8 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 256"
H="python bench.py --steps 8 --warmup 3 --no-walabot --no-u8 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 256"
for rep in 1 2 3 4 5 6; do for ch in 0 12288 16384; do
  if [ $ch = 0 ]; then unset RML_CHUNK; else export RML_CHUNK=$ch; fi
  timeout 900 $W > $O/w${ch}_$rep.json 2>> $O/b.err
  echo -n "walabot ch$ch: "; python tools/exp/show_bench.py $O/w${ch}_$rep.json x | grep "derive" | cut -c1-60
done; done
for rep in 1 2 3; do for ch in 0 12288 16384; do
  if [ $ch = 0 ]; then unset RML_CHUNK; else export RML_CHUNK=$ch; fi
  timeout 900 $H > $O/h${ch}_$rep.json 2>> $O/b.err
  echo -n "64x64x128 ch$ch: "; python tools/exp/show_bench.py $O/h${ch}_$rep.json x | grep "derive" | cut -c1-60
done; done
