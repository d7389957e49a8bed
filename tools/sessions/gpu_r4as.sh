#!/bin/bash
# round-4 GPU session AS: two or three workspaces at 64x64x128, five interleaved rounds (the timeline shows the projection stream idle ~60 us per chunk waiting for its workspace)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4as; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 8 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --no-slice --no-walabot --parity 512"
for rep in 1 2 3 4 5; do for nb in 2 3; do
  RML_NBUF=$nb timeout 600 $B > $O/nb${nb}_$rep.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/nb${nb}_$rep.json nbuf$nb | grep headline | cut -c1-90
done; done
