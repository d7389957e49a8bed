#!/bin/bash
# round-4 GPU session K: kernel timeline of the derive -> slice -> SVM pipeline at the Walabot grid, both pairings
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4k; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for v in 1 0; do
  RML_DERIVE_PIPE=$v python tools/slice_pipe.py
  cd /tmp
  RML_DERIVE_PIPE=$v timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof$v -o k -- python $R/tools/slice_pipe.py --steps 2 > /dev/null 2> $R/$O/prof$v.err
  cd $R
  python tools/timeline.py $(find $O/prof$v -name "*.db" | head -1) --match k_derive_slice --rows 36 > $O/timeline_derive_pipe$v.txt 2>&1
  rm -rf $O/prof$v
  cut -c1-150 $O/timeline_derive_pipe$v.txt
done
