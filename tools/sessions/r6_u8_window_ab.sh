#!/bin/bash
# A/B of the rolling load window in k_project_u8_max (session r6e): libradarml_hip_base.so = the library with round 5's loop
# (NM loads, wait, reduce).  Alternating processes on one box, both grids: the kernel alone (codes + statistics) and the uint8
# pipeline rows of bench.py.
cd $(dirname $0)/../..
for rep in 1 2; do
  for lib in base new; do
    if [ $lib = base ]; then export RML_LIB=$PWD/radar-ml_amd/libradarml_hip_base.so; else unset RML_LIB; fi
    for g in "64x64x128 32768" "22x31x176 131072"; do
      set -- $g
      python tools/kbench.py proj --u8 --grid $1 --frames $2 --iters 15 2>/dev/null | grep "codes+stats only\|f32 rows /255" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('rep $rep lib $lib $1 kernel alone %-30s ms_med %.4f  frac %.4f' % (d['what'][:30], d['ms_med'], d['frac_of_8TBs']))"
    done
  done
done
for rep in 1 2; do
  for lib in base new; do
    if [ $lib = base ]; then export RML_LIB=$PWD/radar-ml_amd/libradarml_hip_base.so; else unset RML_LIB; fi
    python bench.py --no-cpu --no-pmc --parity 256 --steps 5 --warmup 2 --no-slice --no-general --no-dnn --no-sgan 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())['summary']
for g, r in d.items():
    if isinstance(r, dict) and 'u8' in r:
        print('rep $rep lib $lib pipeline %s: u8 v %.0f e2e %.4f roof %.4f same_bits %s | f32 v %.0f' % (g, r['u8']['v'], r['u8']['e2e'], r['u8']['roof'], r['u8']['same_bits'], r['f32']['v']))"
  done
done
