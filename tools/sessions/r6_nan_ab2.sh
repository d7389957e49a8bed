#!/bin/bash
# Session r6k: the headline pipeline, three libraries alternating on one box: this tree (v_maximum3_f32), this tree with maxNum back
# (libradarml_hip_maxnum.so: project*.hip rebuilt with v_max_f32 / fmaxf, everything else the same objects), the commit before.
R=$PWD
OUT=$R/gpurun_out/r6k
mkdir -p $OUT
for round in 1 2 3; do
  for arm in new maxnum prev; do
    case $arm in new) unset RML_LIB;; maxnum) export RML_LIB=$R/radar-ml_amd/libradarml_hip_maxnum.so;; prev) export RML_LIB=$R/radar-ml_amd/libradarml_hip_prev.so;; esac
    echo "## round $round arm $arm" >> $OUT/bench.log
    python bench.py --no-dnn --no-sgan --no-general --no-u8 --no-slice --no-cpu --no-pmc --no-walabot --steps 10 --warmup 3 --parity 512 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['summary']
print(json.dumps({k:(v['f32']['v'],v['f32']['e2e'],v['f32']['roof'],v['f32'].get('gemm')) for k,v in s.items() if isinstance(v,dict) and 'f32' in v}))" >> $OUT/bench.log
  done
done
cat $OUT/bench.log
