#!/bin/bash
# round-4 GPU session J: kernel timeline of the two-stream CNN path
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4j; mkdir -p $O
export TMPDIR=/tmp
cat > /tmp/dnn_tl.py <<'PY'
import sys, importlib
sys.path.insert(0, sys.argv[1])
import torch, radar_ml_amd as rml
dnn = importlib.import_module("radar_ml_amd.dnn")
torch.manual_seed(1)
m = dnn.define_classifier(device="cuda").eval()
V, _ = rml.synth_volumes(49152, 22, 31, 176, seed=5)
for _ in range(2): m.predict_volumes(V, overlap=True)
torch.cuda.synchronize()
PY
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof -o k -- python /tmp/dnn_tl.py $R > /dev/null 2> $R/$O/prof.err
cd $R
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/r4j/prof/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
rows = c.execute("select s.kernel_name, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
rows = [r for r in rows if not r[0].startswith("k_synth")]
t0 = rows[len(rows) // 2][1]
half = rows[len(rows) // 2: len(rows) // 2 + 40]
for name, st, en, q in half:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    print("%9.1f %9.1f %8.1f us  q%-3s %s" % ((st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, q, short))
PY
rm -rf $O/prof
