#!/bin/bash
# A/B of the explicit vmcnt drains in k_derive_slice (session r6c): libradarml_hip_base.so = the library without them.
# Alternating processes on one box, both grids, derive only / derive -> slice.
cd $(dirname $0)/../..
for rep in 1 2; do
  for lib in base new; do
    if [ $lib = base ]; then export RML_LIB=$PWD/radar-ml_amd/libradarml_hip_base.so; else unset RML_LIB; fi
    for g in "64x64x128 16384" "22x31x176 65536"; do
      set -- $g
      echo "== rep $rep lib $lib grid $1"
      python tools/kbench.py derive --grid $1 --frames $2 --iters 15 | grep "(k_derive_slice)" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %-40s ms_med %.4f min %.4f  frac %.4f' % (d['what'][:40], d['ms_med'], d['ms_min'], d['frac_of_8TBs']))"
    done
  done
done
