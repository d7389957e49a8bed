#!/bin/bash
# round-4 GPU session BL: k_c1_apply_pad_pk (packed first-layer forward) -- tests, step time against RML_C1_PK=0, kernel times; derive chunk 12288 at the Walabot grid
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4bl; mkdir -p $O
R=$PWD
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_nn_gpu.py -x -q -k "sgan or bn_lrelu or conv1" 2>&1 | tail -n 3
for k in 0 1 0 1; do echo -n "RML_C1_PK=$k "; RML_C1_PK=$k timeout 600 python tools/bench_nn.py sgan --steps 100 2>&1 | tail -n 1 | cut -c60-200; done
timeout 900 python -m pytest tests/test_svm_gpu.py -x -q -k "slice or derive or read_compare" 2>&1 | tail -n 2
W="python bench.py --steps 8 --warmup 3 --grid 22x31x176 --frames 262144 --no-walabot --no-u8 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 256"
for rep in 1 2 3; do for ch in 0 8192; do
  if [ $ch = 0 ]; then unset RML_CHUNK; else export RML_CHUNK=$ch; fi
  timeout 900 $W > $O/w${ch}_$rep.json 2>> $O/b.err
  echo -n "walabot ch$ch: "; python tools/exp/show_bench.py $O/w${ch}_$rep.json x | grep "derive" | cut -c1-60
done; done
unset RML_CHUNK
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/tools/bench_nn.py sgan --steps 40 > $O/run.log 2>&1
python $R/tools/prof_summary.py stats $(find $O/prof -name "*.db" | head -1) > $O/sgan_stats.txt 2>&1; grep -E "k_c1_" $O/sgan_stats.txt | cut -c1-50,95-150
rm -rf $O/prof
