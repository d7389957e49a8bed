#!/bin/bash
# round-4 GPU session M: resize with double-staged images (one cvt per pixel instead of one per tap)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4m; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_nn_gpu.py -x -q -k "dnn or resize or preprocess" 2>&1 | tail -n 6
python - <<'PY'
import sys, time, importlib, os
sys.path.insert(0, ".")
import torch, radar_ml_amd as rml
nc = importlib.import_module("radar_ml_amd.nn_common")
V, _ = rml.synth_volumes(8192, 22, 31, 176, seed=5)
feat = rml.process_volumes(V, mode="max", scale=False)
_, q, _, _, _ = rml.process_volumes(V, mode="max", scale=False, codes="only")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for dbl in ("1", "0"):
    os.environ["RML_RESIZE_DBL"] = dbl
print("resize 3 planes of 8192 frames: float rows %.3f ms, codes %.3f ms (double staging)" % (
    t(lambda: nc.preprocess_features(feat, (22, 31, 176), (80, 80), out_dtype="bfloat16")),
    t(lambda: nc.preprocess_codes(q, (22, 31, 176), (80, 80), out_dtype="bfloat16"))))
dnn = importlib.import_module("radar_ml_amd.dnn")
torch.manual_seed(1)
m = dnn.define_classifier(device="cuda").eval()
V, _ = rml.synth_volumes(65536, 22, 31, 176, seed=5)
for tag, vol in (("f32", V), ("u8", V.to(torch.uint8))):
    for cd in (True, False):
        for _ in range(2): m.predict_volumes(vol, codes=cd)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): p = m.predict_volumes(vol, codes=cd)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("dnn %s codes=%s: %.2f M frames/s, %.3f ms per 8192" % (tag, cd, 65536 / dt / 1e6, dt / 8 * 1e3))
PY
RML_RESIZE_DBL=0 python - <<'PY'
import sys, time, importlib
sys.path.insert(0, ".")
import torch, radar_ml_amd as rml
nc = importlib.import_module("radar_ml_amd.nn_common")
V, _ = rml.synth_volumes(8192, 22, 31, 176, seed=5)
feat = rml.process_volumes(V, mode="max", scale=False)
_, q, _, _, _ = rml.process_volumes(V, mode="max", scale=False, codes="only")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("RML_RESIZE_DBL=0: float rows %.3f ms, codes %.3f ms (float staging)" % (
    t(lambda: nc.preprocess_features(feat, (22, 31, 176), (80, 80), out_dtype="bfloat16")),
    t(lambda: nc.preprocess_codes(q, (22, 31, 176), (80, 80), out_dtype="bfloat16"))))
PY
