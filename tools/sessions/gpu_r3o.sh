#!/bin/bash
# round-3 GPU session O: the full -m gpu suite, the per-workload profile round, the full default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3o; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest_gpu.txt
timeout 1500 bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/pytest_gpu.txt; tail -40 $O/profile_round.log | cut -c1-200; python tools/exp/show_bench.py $O/bench.json full; tail -3 $O/bench.err
