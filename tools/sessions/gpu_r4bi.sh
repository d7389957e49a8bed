#!/bin/bash
# round-4 GPU session BI: new conv1 tests; chunk size of the derive -> slice -> SVM pairing at the Walabot grid (RML_CHUNK), two interleaved rounds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4bi; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_nn_gpu.py -x -q -k "conv1" 2>&1 | tail -n 3
B="python bench.py --steps 8 --warmup 3 --grid 22x31x176 --frames 262144 --no-walabot --no-u8 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 256"
for rep in 1 2; do for ch in 0 12288 16384 24576; do
  if [ $ch = 0 ]; then unset RML_CHUNK; else export RML_CHUNK=$ch; fi
  timeout 900 $B > $O/ch${ch}_$rep.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/ch${ch}_$rep.json ch$ch | grep "derive\|headline" | cut -c1-125
done; done
