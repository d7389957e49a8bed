#!/bin/bash
# round-4 GPU session AP: chunk size of the slice_mode row (ijk given): whole-round chunks of the ring GEMM give 2-3 chunks per 65 536 frames
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4ap; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 6 --warmup 2 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --parity 512"
run() { tag=$1; shift; env "$@" timeout 900 $B > $O/$tag.json 2>> $O/b.err; python tools/exp/show_bench.py $O/$tag.json $tag | grep "slice_mode\|derive" | cut -c1-150; }
run base RML_X=0
run c8k RML_CHUNK=8192
run c16k RML_CHUNK=16384
