#!/usr/bin/env python3
"""Why did predict_volumes(V) twice not give the same bits (test_dnn_full_size..., session r6a)?  The same sequence of calls as the
test, last_guard after each, and the rows / stages that differ between the first and the last call."""
import copy
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import radar_ml_amd as rml
import test_nn_gpu as T

dnn = importlib.import_module("radar_ml_amd.dnn")
cpu, _, _ = T._train_classifier_with_margins(dnn)
gpu = copy.deepcopy(cpu).to("cuda").eval()
frames = 32768
V, _ = rml.synth_volumes(frames, 22, 31, 176, seed=31)


def call(tag, *a, **k):
    p = gpu.predict_volumes(*a, **k)
    print(tag, {kk: (round(v, 6) if isinstance(v, float) else v) for kk, v in gpu.last_guard.items()}, flush=True)
    return p


whole = call("whole#1", V)
again = call("whole#2", V)
print("  #1 == #2:", torch.equal(whole, again), "rows differing:", int((whole != again).any(1).sum()), "max", float((whole - again).abs().max()))
cuts = [0, frames // 3 + 777, frames // 3 + 777 + 9999, frames]
for lo, hi in zip(cuts[:-1], cuts[1:]):
    call("cut %d:%d" % (lo, hi), V[lo:hi])
call("5000/1024", V[:5000], batch_size=1024)
call("first 16384", V[:16384])
third = call("whole#3", V)
d = (whole != third).any(1)
print("  #1 == #3:", torch.equal(whole, third), "rows differing:", int(d.sum()), "max", float((whole - third).abs().max()))
print("  #2 == #3:", torch.equal(again, third), "rows differing:", int((again != third).any(1).sum()))
raw = gpu.predict_volumes(V, label_guard=None)
g = dnn.Classifier._gaps(raw)
if int(d.sum()):
    idx = d.nonzero().squeeze(1)[:16]
    print("  bf16 gaps of differing rows:", [round(float(x), 5) for x in g[idx]])
    print("  |#1-#3| there:", [float(x) for x in (whole[idx] - third[idx]).abs().max(1).values])
# is the x3 re-scoring itself composition-independent?  the same rows alone and inside a larger index set
rows = (g < 2e-2).nonzero().squeeze(1)
print("candidates at 2e-2:", int(rows.numel()))
a = gpu.rescore_exact(V, precision="x3", rows=rows)
b = gpu.rescore_exact(V, precision="x3", rows=rows[: rows.numel() // 2])
print("  x3 rescoring: first half alone == inside the whole set:", torch.equal(a[: rows.numel() // 2], b), float((a[: rows.numel() // 2] - b).abs().max()))
fa = gpu.exact_features(V, rows)
fb = gpu.exact_features(V, rows[: rows.numel() // 2])
print("  x3 features: ", torch.equal(fa[: rows.numel() // 2], fb))
