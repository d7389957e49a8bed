#!/bin/bash
# A/B of the rolling load window in k_project_fast (session r6aa): libradarml_hip_base.so = the library with the `#pragma unroll 2`
# loop that hipcc never unrolled.  configs[1] (projection only, float rows) at 4 096 / 16 384 frames, codes only, uint8 volumes
# through the float kernel's <uint8_t> instantiation are not affected (k_project_u8_max takes them).
cd $(dirname $0)/../..
for rep in 1 2 3; do
  for lib in base new; do
    if [ $lib = base ]; then export RML_LIB=$PWD/radar-ml_amd/libradarml_hip_base.so; else unset RML_LIB; fi
    for fr in 4096 16384; do
      python tools/kbench.py proj --grid 64x64x128 --frames $fr --iters 15 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('rep $rep lib $lib B $fr %-34s ms_med %.4f  frac %.4f' % (d['what'][:34], d['ms_med'], d['frac_of_8TBs']))"
    done
  done
done
