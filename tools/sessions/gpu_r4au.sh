#!/bin/bash
# round-4 GPU session AU: the whole GPU suite and the smoke test on the final tree
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -n 7
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
