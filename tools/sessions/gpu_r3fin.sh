#!/bin/bash
# round-3 final GPU session: the full -m gpu suite, the per-workload profile round, the full default bench line, smoke()
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3fin; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 5 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
timeout 1500 bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python tools/exp/show_bench.py $O/bench.json full; tail -n 3 $O/bench.err
