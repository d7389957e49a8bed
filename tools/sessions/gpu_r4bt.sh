#!/bin/bash
# round-4 GPU session BT: the closing run on the final tree -- whole GPU suite, smoke, the round's profile set, the default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4bt; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( time timeout 2400 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -n 7
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
( time timeout 2400 bash tools/profile_round.sh r04 ) 2>&1 | tail -n 5
cd $R
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python tools/exp/show_bench.py $O/bench.json final | cut -c1-150
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4bt/bench.json"))
print(json.dumps(d["summary"]["dnn_configs3"]), json.dumps(d["summary"]["sgan_configs4"]), json.dumps(d["summary"]["proj_only_configs1"]))
PY
