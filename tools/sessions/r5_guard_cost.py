"""Where the margin guard's time goes (session r5d): pieces of Classifier._guard timed with synchronisation between them."""
import importlib, sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import radar_ml_amd as rml
dnn = importlib.import_module("radar_ml_amd.dnn")
common = importlib.import_module("radar_ml_amd.common")
nn_common = importlib.import_module("radar_ml_amd.nn_common")
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = dnn.define_classifier(device=dev).eval()
V, _ = rml.synth_volumes(65536, 22, 31, 176, seed=5, device=dev)

def T(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r

ms, p = T(lambda: m.predict_volumes(V, label_guard=None)); print("chain without guard %.2f ms" % ms)
for k in (64, 577, 4096):
    idx = torch.arange(0, 65536, 65536 // k, device=dev)[:k]
    ms, _ = T(lambda: p.float().topk(2, dim=1).values); print("k=%d topk %.3f ms" % (k, ms))
    ms, _ = T(lambda: ((p[:, 0] - p[:, 1]).abs() < 0.5).nonzero()); print("   nonzero %.3f ms" % ms)
    ms, vs = T(lambda: V[idx]); print("   gather volumes %.3f ms" % ms)
    ms, feat = T(lambda: common.process_volumes(vs, mode="max", scale=False)); print("   process_volumes %.3f ms" % ms)
    ms, xs = T(lambda: nn_common.preprocess_features(feat, (22, 31, 176), (80, 80), out_dtype="float32")); print("   exact resize %.3f ms" % ms)
    x4 = [x.reshape(x.shape[0], 1, 80, 80) for x in xs]
    ms, _ = T(lambda: m(*x4)); print("   float32 torch layers %.3f ms" % ms)
    ms, _ = T(lambda: m.forward_float64(*[x[:8] for x in xs])); print("   float64 layers, 8 rows %.3f ms" % ms)
    with torch.no_grad():
        ms, _ = T(lambda: m.rescore_exact(vs, (80, 80), "max", "float32")); print("   rescore_exact f32 total %.3f ms" % ms)
ms, _ = T(lambda: m.predict_volumes(V)); print("chain with guard %.2f ms, %s" % (ms, m.last_guard))
