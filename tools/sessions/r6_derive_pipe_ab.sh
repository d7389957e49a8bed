#!/bin/bash
# the derive -> slice -> SVM pipeline rows of bench.py with and without the explicit vmcnt drains of k_derive_slice (session r6d)
cd $(dirname $0)/../..
for rep in 1 2; do
  for lib in base new; do
    if [ $lib = base ]; then export RML_LIB=$PWD/radar-ml_amd/libradarml_hip_base.so; else unset RML_LIB; fi
    for g in "64x64x128 65536" "22x31x176 262144"; do
      set -- $g
      python bench.py --no-cpu --no-pmc --parity 256 --steps 5 --warmup 2 --grid $1 --frames $2 --no-walabot --no-u8 --no-general --no-dnn --no-sgan 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())['summary']
for g, r in d.items():
    if isinstance(r, dict) and 'derive_slice_svm' in r:
        x = r['derive_slice_svm']; f = r['f32']
        print('rep $rep lib $lib %s: derive_slice_svm v %.0f e2e %.4f roof %.4f | max f32 v %.0f e2e %.4f' % (g, x['v'], x['e2e'], x['roof'], f['v'], f['e2e']))"
    done
  done
done
