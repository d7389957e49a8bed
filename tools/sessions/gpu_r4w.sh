#!/bin/bash
# round-4 GPU session W: fc1 v4 (lane-contiguous loads, register prefetch, LDS stages), row-major and K-block layouts
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4w; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_nn_gpu.py -x -q -k "preprocess or predict_volumes or dnn or dense or kblock" 2>&1 | tail -n 8
for rep in 1 2; do
  timeout 300 python tools/dnn_chain.py
  timeout 300 python tools/dnn_chain.py --u8
done
R=$PWD
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o dnn -- python $R/tools/dnn_chain.py --steps 3 > $R/$O/prof.log 2>&1; cd $R
python tools/prof_summary.py stats $O/prof/dnn_results.db > $O/stats_dnn_chain.txt; head -9 $O/stats_dnn_chain.txt | cut -c1-150
rm -rf $O/prof
python - <<'PY'
import importlib, torch, time
dnn = importlib.import_module("radar_ml_amd.dnn")
m = dnn.define_classifier(device="cuda").eval()
fv = torch.relu(torch.randn((8192, 38400), device="cuda")).to(torch.bfloat16)
kb = torch.relu(torch.randn((600, 8192, 64), device="cuda")).to(torch.bfloat16)
for name, fn in (("rows", lambda: m.dense_tail(fv)), ("kblock", lambda: m.dense_tail(kb, kblock=True)), ("hipblaslt", lambda: m.dense_tail(fv, fused=False))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); print("dense tail %-10s %.1f us" % (name, (time.perf_counter() - t0) / 20 * 1e6))
PY
