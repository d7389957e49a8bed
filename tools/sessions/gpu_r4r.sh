#!/bin/bash
# round-4 GPU session R (re-entry): whole GPU suite + smoke + default bench line on the restored tree
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4r; mkdir -p $O
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -n 6 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 3
( time timeout 1200 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2500 $O/bench.json
