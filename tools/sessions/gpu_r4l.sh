#!/bin/bash
# round-4 GPU session L: CNN path with the uint8-code hand-over between projection and resize, cached bf16 tail weights
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4l; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_nn_gpu.py -x -q -k "dnn or resize or preprocess" 2>&1 | tail -n 6
python - <<'PY'
import sys, time, importlib
sys.path.insert(0, ".")
import torch, radar_ml_amd as rml
dnn = importlib.import_module("radar_ml_amd.dnn")
torch.manual_seed(1)
m = dnn.define_classifier(device="cuda").eval()
V, _ = rml.synth_volumes(65536, 22, 31, 176, seed=5)
for tag, vol in (("f32", V), ("u8", V.to(torch.uint8))):
    for cd in (True, False, True, False):
        for _ in range(2): m.predict_volumes(vol, codes=cd)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): p = m.predict_volumes(vol, codes=cd)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("dnn %s codes=%s: %.2f M frames/s, %.3f ms per 8192" % (tag, cd, 65536 / dt / 1e6, dt / 8 * 1e3))
PY
R=$PWD
cat > /tmp/dnn_tl.py <<'PY'
import sys, importlib
sys.path.insert(0, sys.argv[1])
import torch, radar_ml_amd as rml
dnn = importlib.import_module("radar_ml_amd.dnn")
torch.manual_seed(1)
m = dnn.define_classifier(device="cuda").eval()
V, _ = rml.synth_volumes(65536, 22, 31, 176, seed=5)
for _ in range(3): m.predict_volumes(V)
torch.cuda.synchronize()
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o k -- python /tmp/dnn_tl.py $R > /dev/null 2> $R/$O/prof.err
cd $R
python tools/prof_summary.py stats $(find $O/prof -name "*.db" | head -1) | head -24 | cut -c1-150
rm -rf $O/prof
