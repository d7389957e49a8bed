#!/bin/bash
# round-4 GPU session AG: pipeline knobs again with the code stage in place (chunk size, third workspace, split streams), Walabot grid and headline
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4ag; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 8 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --no-slice --parity 1024"
run() { tag=$1; shift; env "$@" timeout 900 $B > $O/$tag.json 2>> $O/b.err; python tools/exp/show_bench.py $O/$tag.json $tag | grep -v gate | cut -c1-120; }
for rep in 1 2; do
  run base_$rep RML_X=0
  run chunk12k_$rep RML_CHUNK=12288
  run chunk16k_$rep RML_CHUNK=16384
  run nbuf3_$rep RML_NBUF=3
  run split_$rep RML_PIPE_SPLIT=1 RML_NBUF=3
done
