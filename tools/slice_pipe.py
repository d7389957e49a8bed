#!/usr/bin/env python3
"""The derive -> slice -> SVM front door alone (no fit: a synthetic exact model), for timelines and A/B runs of its pipeline:

    python tools/slice_pipe.py [--grid 22x31x176] [--frames 131072] [--svs 2281] [--steps 4] [--mode slice|max]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="22x31x176")
    ap.add_argument("--frames", type=int, default=131072)
    ap.add_argument("--svs", type=int, default=2281)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--mode", default="slice")
    ap.add_argument("--u8", action="store_true")
    a = ap.parse_args()
    import torch
    import radar_ml_amd as rml
    X, Y, Z = (int(t) for t in a.grid.split("x"))
    M = a.svs
    svv, _ = rml.synth_volumes(M, X, Y, Z, seed=101)
    sv = rml.process_volumes(svv, mode="max", scale=True).cpu().numpy().astype(np.float64)
    rng = np.random.default_rng(0)
    ns = np.array([M // 3, M // 3, M - 2 * (M // 3)], dtype=np.int32)
    svc = rml.GpuSVC(sv, rng.uniform(-1, 1, (2, M)), rng.uniform(-0.1, 0.1, 3), ns, 0.01, np.arange(3),
                     calib_a=-np.ones(3), calib_b=np.zeros(3))
    V, _ = rml.synth_volumes(a.frames, X, Y, Z, seed=7)
    if a.u8:
        V = V.to(torch.uint8)
    fn = lambda: svc.decide_volumes(V, mode=a.mode, scale=True, want_proba=True)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    fb = (1 if a.u8 else 4) * X * Y * Z + 16
    print(json.dumps({"mode": a.mode, "grid": [X, Y, Z], "frames": a.frames, "ms": round(dt * 1e3, 3), "frames_per_s": round(a.frames / dt),
                      "e2e_frac_of_8TBs": round(a.frames / dt * fb / 8e12, 4)}))


if __name__ == "__main__":
    main()
