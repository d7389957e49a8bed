#!/bin/bash
# round-4 GPU session AH: three workspaces by default for large float32 frames -- headline incl. its derive / slice rows, A/B against RML_NBUF=2
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4ah; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_svm_gpu.py -x -q -k "full_size or chunk or pipeline or volumes" 2>&1 | tail -n 3
B="python bench.py --steps 8 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --no-walabot --parity 1024"
run() { tag=$1; shift; env "$@" timeout 900 $B > $O/$tag.json 2>> $O/b.err; python tools/exp/show_bench.py $O/$tag.json $tag | grep -v gate | cut -c1-130; }
for rep in 1 2; do
  run nbuf2_$rep RML_NBUF=2
  run default_$rep RML_X=0
done
