#!/bin/bash
# round-4 GPU session S: fused CNN preprocessing (csrc/preprocess.hip) -- tests, chain A/B, kernel summary
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4s; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nn_gpu.py -x -q -k "preprocess or predict_volumes or dnn" 2>&1 | tail -n 15
for rep in 1 2; do
  timeout 300 python tools/dnn_chain.py --exact
  timeout 300 python tools/dnn_chain.py
  timeout 300 python tools/dnn_chain.py --u8
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o dnn -- python $OLDPWD/tools/dnn_chain.py --steps 3 > $OLDPWD/$O/prof.log 2>&1; cd $OLDPWD
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r4s/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print("%-70s calls %6s avg %10.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
