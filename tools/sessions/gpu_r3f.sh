#!/bin/bash
# round-3 GPU session F: whole GPU suite + the bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3f; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.txt
timeout 1500 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
cat $O/pytest_gpu.txt; tail -n 5 $O/bench.err; python tools/exp/show_bench.py $O/bench.json 2>/dev/null | head -80
