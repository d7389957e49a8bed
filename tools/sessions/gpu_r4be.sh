#!/bin/bash
# round-4 GPU session BE: this tree's library against the one of commit c3ef2bc (before read-compare-write) on ONE box, three interleaved rounds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4be; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 8 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 1024"
for rep in 1 2 3; do for k in old new new0; do
  unset RML_LIB RML_CODE_RMW
  if [ $k = old ]; then export RML_LIB=$PWD/radar-ml_amd/libradarml_hip_c3ef2bc.so; fi
  if [ $k = new0 ]; then export RML_CODE_RMW=0; fi
  timeout 900 $B > $O/${k}_$rep.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/${k}_$rep.json $k | grep -v "gate\|slice_mode" | cut -c1-125
done; done
