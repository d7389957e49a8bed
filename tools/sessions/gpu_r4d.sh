#!/bin/bash
# round-4 GPU session D: the new bench line (slice rows, parity gate, compact tail) at reduced and full size
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --steps 2 --warmup 1 --frames 4096 --walabot-frames 8192 --train 1500 --parity 256 --general-frames 512 --dnn-frames 1024 --dnn-parity 64 > $O/small.json 2> $O/small.err
echo "small rc=$?"; tail -c 2500 $O/small.json; echo; tail -n 5 $O/small.err
python tools/exp/show_bench.py $O/small.json small
timeout 1500 python bench.py --steps 5 --warmup 2 --no-general --no-dnn --no-sgan > $O/full.json 2> $O/full.err
echo "full rc=$?"; tail -c 2200 $O/full.json; echo; tail -n 5 $O/full.err
python tools/exp/show_bench.py $O/full.json full
