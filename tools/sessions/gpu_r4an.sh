#!/bin/bash
# round-4 GPU session AN: the default bench line on the final library (tracked as profiles/r04_bench_1gpu.json), CNN chain kernel summary
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4an; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python tools/exp/show_bench.py $O/bench.json final | cut -c1-150
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4an/bench.json"))
print(json.dumps(d["summary"]["dnn_configs3"]), json.dumps(d["summary"]["sgan_configs4"]), json.dumps(d["summary"]["proj_only_configs1"]))
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o dnn -- python $R/tools/dnn_chain.py --steps 3 > $R/$O/prof.log 2>&1; cd $R
python tools/prof_summary.py stats $O/prof/dnn_results.db > $O/stats_dnn_chain.txt; head -10 $O/stats_dnn_chain.txt | cut -c1-150
rm -rf $O/prof
