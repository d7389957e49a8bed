#!/usr/bin/env python3
"""BASELINE configs[3] chain alone (Classifier.predict_volumes at the Walabot grid), for kernel profiles and A/B runs:

    python tools/dnn_chain.py [--frames 65536] [--steps 5] [--exact] [--u8]

prints frames/s; under `rocprofv3 --kernel-trace --stats` the per-kernel summary of the chain."""
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
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--exact", action="store_true", help="float rows + the Pillow-bit-identical resize (the round-1..3 chain)")
    ap.add_argument("--u8", action="store_true")
    a = ap.parse_args()
    import torch
    import radar_ml_amd as rml
    dnn = importlib.import_module("radar_ml_amd.dnn")
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    model = dnn.define_classifier(device=dev).eval()
    V, _ = rml.synth_volumes(a.frames, 22, 31, 176, seed=5)
    if a.u8:
        V = V.to(torch.uint8)
    for _ in range(2):
        p = model.predict_volumes(V, batch_size=a.batch, exact_resize=a.exact)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        p = model.predict_volumes(V, batch_size=a.batch, exact_resize=a.exact)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print("dnn chain %s%s: %d frames in %.3f ms = %.3f M frames/s" % ("exact-resize" if a.exact else "fused-preprocess", " u8" if a.u8 else "",
                                                                       a.frames, dt * 1e3, a.frames / dt / 1e6))


if __name__ == "__main__":
    main()
