#!/usr/bin/env python3
"""k_pre3 alone (code rows -> three 80x80 bf16 planes) at the Walabot grid: python tools/pre3_bench.py [--frames 16384]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=16384); a = ap.parse_args()
import importlib, torch
import radar_ml_amd as rml
nc = importlib.import_module("radar_ml_amd.nn_common")
grid = (22, 31, 176)
V, _ = rml.synth_volumes(a.frames, *grid, seed=3)
_, q, isum, isq, flags = rml.process_volumes(V, mode="max", codes=True)
for _ in range(3): nc.preprocess_rows(grid, (80, 80), codes=q)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): nc.preprocess_rows(grid, (80, 80), codes=q)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print("k_pre3: %d samples in %.1f us = %.2f TB/s of (10 010 + 38 400) B per sample" % (a.frames, dt * 1e6, a.frames * 48410 / dt / 1e12))
