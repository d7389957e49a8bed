#!/bin/bash
# round-4 GPU session BR: WRITE_SIZE of the pipeline's projection kernels with and without read-compare-write of the code rows (64x64x128 uint8 and derive -> slice), kernel trace + one counter only
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4br; mkdir -p $O
R=$PWD
export TMPDIR=/tmp
cd /tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-walabot --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 256"
for k in 0 d; do
  if [ $k = 0 ]; then export RML_CODE_RMW=0; else unset RML_CODE_RMW; fi
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace -d $O/p_${k}_$c -o k -- $B > $O/b_${k}_$c.json 2> $O/b_${k}_$c.err
    echo "== RML_CODE_RMW=$k $c"; python $R/tools/pmc_query.py $(find $O/p_${k}_$c -name "*.db" | head -1) "%k_project_u8%" | cut -c1-140
    python $R/tools/pmc_query.py $(find $O/p_${k}_$c -name "*.db" | head -1) "%k_derive_slice%" | cut -c1-140
    rm -rf $O/p_${k}_$c
  done
done > $O/rmw_pmc.txt 2>&1
cat $O/rmw_pmc.txt
