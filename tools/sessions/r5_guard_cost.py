"""Where the margin guard's time goes (sessions r5d-r5f): pieces of Classifier._guard timed with synchronisation between them."""
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

with torch.no_grad():
    ms, p = T(lambda: m.predict_volumes(V, label_guard=None)); print("chain without guard %.2f ms" % ms)
    for k in (32, 256):
        idx = torch.arange(0, 65536, 65536 // k, device=dev)[:k]
        vs = V[idx]
        feat = common.process_volumes(vs, mode="max", scale=False)
        xs = nn_common.preprocess_features(feat, (22, 31, 176), (80, 80), out_dtype="float32")
        for prec in ("float32", "float64"):
            ms, _ = T(lambda: m.forward_exact(*xs, precision=prec)); print("k=%d forward_exact %s %.3f ms" % (k, prec, ms))
            ms, _ = T(lambda: m.rescore_exact(vs, (80, 80), "max", prec)); print("      rescore_exact %s total %.3f ms" % (prec, ms))
    # a guard that flags ~1 % of the rows: threshold chosen on this model's own gap distribution
    t2 = p.topk(2, dim=1).values
    gap = float(torch.quantile((t2[:, 0] - t2[:, 1])[:16384], 0.01))
    ms, _ = T(lambda: m.predict_volumes(V, label_guard=gap)); print("chain with a guard at the 1 %% quantile (%.2e): %.2f ms, %s" % (gap, ms, m.last_guard))
    # configs[1] stand-alone projection, float32 rows out (the row that read 0.68 instead of 0.78 in session r5d)
    V2, _ = rml.synth_volumes(16384, 64, 64, 128, seed=5, device=dev)
    out = torch.empty((16384, 20480), device=dev)
    for nb in (4096, 16384, 4096, 16384):
        ms, _ = T(lambda: rml.process_volumes(V2[:nb], mode="max", out=out[:nb]), n=10)
        print("process_volumes 64x64x128 batch %d: %.3f ms = %.3f of 8 TB/s" % (nb, ms, nb * (2097152 + 81920) / (ms * 1e-3) / 8e12))
