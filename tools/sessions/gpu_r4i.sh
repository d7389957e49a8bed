#!/bin/bash
# round-4 GPU session I: generalised k_project_lin (tests + alone rates), CNN row on two streams (new schedule) vs one, tests touched by the clean-up
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4i; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_projection_gpu.py tests/test_svm_gpu.py tests/test_nn_gpu.py -x -q -k "dnn" 2>&1 | tail -n 6
for g in ; do for lin in 1 0; do
  RML_LINPLANE=$lin timeout 300 python tools/kbench.py proj --grid $g --frames 16384 --iters 10 2>&1 | grep "codes+stats" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   lin=$lin $g', d['what'], d['ms_med'], d['frac_of_8TBs'])"
done; done
python - <<'PY'
import sys, time, importlib
sys.path.insert(0, ".")
import torch, radar_ml_amd as rml
dnn = importlib.import_module("radar_ml_amd.dnn")
torch.manual_seed(1)
m = dnn.define_classifier(device="cuda").eval()
V, _ = rml.synth_volumes(65536, 22, 31, 176, seed=5)
for tag, vol in (("f32", V), ("u8", V.to(torch.uint8))):
    for ov in (True, False, True, False):
        for _ in range(2): m.predict_volumes(vol, overlap=ov)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): p = m.predict_volumes(vol, overlap=ov)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        ev = []
        m.predict_volumes(vol, overlap=ov, trunk_events=ev); torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b, n in ev]
        print("dnn %s overlap=%s: %.2f M frames/s, %.3f ms per 8192; trunk in situ avg %.3f min %.3f max %.3f ms" % (tag, ov, 65536 / dt / 1e6, dt / 8 * 1e3, sum(ms) / len(ms), min(ms), max(ms)))
PY
