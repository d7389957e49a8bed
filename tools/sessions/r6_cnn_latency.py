#!/usr/bin/env python3
"""One observation through the CNN path (dnn.py:373-381 predicts one target per call): host clock per call, B = 1 and B = 64."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import radar_ml_amd as rml
import bench
dnn = importlib.import_module("radar_ml_amd.dnn")
dev = torch.device("cuda", 0)
model = bench.dnn_bench_model(rml, dev, 1234, 200)
V, _ = rml.synth_volumes(64, 22, 31, 176, seed=5, device=dev)

def timed(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return round(float(np.percentile(np.array(ts) * 1e6, 50)), 1)

print({"predict_volumes B=1 (guard on)": timed(lambda: model.predict_volumes(V[:1])),
       "predict_volumes B=1 (guard off)": timed(lambda: model.predict_volumes(V[:1], label_guard=None)),
       "predict_volumes B=64 (guard on)": timed(lambda: model.predict_volumes(V)),
       "predict_volumes B=64 (guard off)": timed(lambda: model.predict_volumes(V, label_guard=None))})
