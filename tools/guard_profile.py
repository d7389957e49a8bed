#!/usr/bin/env python3
"""The CNN row's margin guard in isolation (radar-ml_amd/dnn.py Classifier._guard): bench.py's model and frames, the call with and
without the guard, and the guard's stages timed one by one with events on the stream they run on.

    python tools/guard_profile.py [--frames 65536] [--steps 5] [--random-init]

Under `rocprofv3 --kernel-trace --stats` the per-kernel summary of the same calls."""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--train-steps", type=int, default=600)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--random-init", action="store_true", help="untrained weights: outputs ~1/3 each, most rows inside the gap")
    a = ap.parse_args()
    import torch
    import radar_ml_amd as rml
    import bench
    dnn = importlib.import_module("radar_ml_amd.dnn")
    nn_common = importlib.import_module("radar_ml_amd.nn_common")
    common = importlib.import_module("radar_ml_amd.common")
    dev = torch.device("cuda", 0)
    if a.random_init:
        torch.manual_seed(a.seed)
        model = dnn.define_classifier(device=dev).eval()
    else:
        model = bench.dnn_bench_model(rml, dev, a.seed, a.train_steps)
    V, _ = rml.synth_volumes(a.frames, 22, 31, 176, seed=a.seed + 7, device=dev)

    def timed(fn, n=a.steps, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, r

    t_on, p = timed(lambda: model.predict_volumes(V))
    g = dict(model.last_guard)
    t_off, raw = timed(lambda: model.predict_volumes(V, label_guard=None))
    print("call: %.3f ms with the guard, %.3f ms without (+%.1f %%): %.2f M frames/s;  last_guard %s" % (t_on, t_off, (t_on / t_off - 1) * 100, a.frames / t_on / 1e3, g))
    # the stages of one round on the rows the guard picked first
    gaps = model._gaps(raw)
    cand = (gaps < dnn.LABEL_GUARD).nonzero().squeeze(1)
    n = int(cand.numel())
    print("stage times for the %d rows below %.0e:" % (n, dnn.LABEL_GUARD))
    if n == 0:
        return
    if n > 16384:
        t, _ = timed(lambda: model.rescore_exact(V, precision="x3", rows=cand), n=2, warm=1); print("  rescore_exact(x3), all %d candidates (dense passes) %8.3f ms" % (n, t))
    cand = cand[:16384]
    n = int(cand.numel())
    t, _ = timed(lambda: model._gaps(raw)); print("  gaps                      %8.3f ms" % t)
    t, _ = timed(lambda: (gaps < dnn.LABEL_GUARD).nonzero()); print("  nonzero (host sync)       %8.3f ms" % t)
    t, v = timed(lambda: V[cand]); print("  gather volumes            %8.3f ms" % t)
    t, feat = timed(lambda: common.process_volumes(v, mode="max", scale=False)); print("  exact projection          %8.3f ms" % t)
    t, xs = timed(lambda: nn_common.preprocess_features(feat, (22, 31, 176), (80, 80), out_dtype="float32")); print("  resize x 3 (Pillow-exact) %8.3f ms" % t)
    t, fv = timed(lambda: model.features_x3(*xs)); print("  x3 trunk                  %8.3f ms  (%.3f us per row)" % (t, t * 1e3 / n))
    t, p3 = timed(lambda: model.forward_exact(*xs, precision="x3")); print("  x3 trunk + float32 tail   %8.3f ms" % t)
    t, _ = timed(lambda: model.rescore_exact(V, precision="x3", rows=cand)); print("  rescore_exact(x3)         %8.3f ms" % t)
    t, _ = timed(lambda: float((raw[cand] - p3).abs().max().cpu())); print("  error read-back           %8.3f ms" % t)
    t, _ = timed(lambda: model.rescore_exact(V, precision="float64", rows=cand[:32])); print("  rescore_exact(float64) 32 %8.3f ms" % t)
    p64 = model.rescore_exact(V, precision="float64", rows=cand[:256])
    p6 = model.rescore_exact(V, precision="x6", rows=cand[:256])
    print("  |x3 - float64| on %d candidate rows: %.2e, |x6 - float64| %.2e   (|bf16 chain - float64|: %.2e)"
          % (min(n, 256), float((p3[:256].double() - p64).abs().max()), float((p6.double() - p64).abs().max()), float((raw[cand[:256]].double() - p64).abs().max())))
    t, _ = timed(lambda: model.rescore_exact(V, precision="x6", rows=cand[:32])); print("  rescore_exact(x6) 32      %8.3f ms" % t)
    t, _ = timed(lambda: model.rescore_exact(V, precision="float32", rows=cand[:1024])); print("  rescore_exact(float32) %4d (round 5's stage) %8.3f ms" % (min(n, 1024), t))


if __name__ == "__main__":
    main()
