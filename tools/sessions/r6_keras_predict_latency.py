#!/usr/bin/env python3
"""Classifier.predict (the Keras surface of dnn.py:373-381: numpy planes in, numpy probabilities out) through the fused chain
against the PyTorch / MIOpen layers under autocast: one target per call and 8 192 per call, host clock."""
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
rng = np.random.default_rng(0)
x = [rng.uniform(-1, 1, (8192, 80, 80, 1)).astype(np.float32) for _ in range(3)]

def timed(fn, n):
    for _ in range(3): fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return round(float(np.percentile(np.array(ts) * 1e6, 50)), 1)

x1 = [a[:1] for a in x]
pf, pa = model.predict(x, fused=True), model.predict(x, fused=False)
print({"N=1 fused us": timed(lambda: model.predict(x1), 100), "N=1 autocast layers us": timed(lambda: model.predict(x1, fused=False), 100),
       "N=8192 fused us": timed(lambda: model.predict(x), 5), "N=8192 autocast layers us": timed(lambda: model.predict(x, fused=False), 5),
       "max |dp| fused vs autocast layers": float(np.abs(pf - pa).max()), "labels equal": bool((pf.argmax(1) == pa.argmax(1)).all())})
