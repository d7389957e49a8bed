#!/bin/bash
# Session r6j: mode MAX with v_maximum3_f32 (np.max's NaN policy) against the library of the commit before (v_max_f32), same box,
# arms alternating: stand-alone projection (codes + statistics, both grids, the pipeline's configuration) and the headline pipeline.
R=$PWD
OUT=$R/gpurun_out/r6j
mkdir -p $OUT
PREV=$R/radar-ml_amd/libradarml_hip_prev.so
for round in 1 2 3; do
  for arm in new prev; do
    if [ $arm = prev ]; then export RML_LIB=$PREV; else unset RML_LIB; fi
    for grid in 64x64x128 22x31x176; do
      fr=8192; [ $grid = 22x31x176 ] && fr=16384
      echo "## round $round arm $arm grid $grid" >> $OUT/kbench.log
      RML_WAVE_SHARE=1 python tools/kbench.py proj --grid $grid --frames $fr --iters 20 2>/dev/null | grep "codes+stats" >> $OUT/kbench.log
    done
  done
done
for round in 1 2; do
  for arm in new prev; do
    if [ $arm = prev ]; then export RML_LIB=$PREV; else unset RML_LIB; fi
    echo "## round $round arm $arm" >> $OUT/bench.log
    python bench.py --no-dnn --no-sgan --no-general --no-u8 --no-slice --no-cpu --no-pmc --steps 10 --warmup 3 --parity 512 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['summary']
print(json.dumps({k:(v['f32']['v'],v['f32']['e2e'],v['f32']['roof']) for k,v in s.items() if isinstance(v,dict) and 'f32' in v}))" >> $OUT/bench.log
  done
done
cat $OUT/kbench.log $OUT/bench.log
