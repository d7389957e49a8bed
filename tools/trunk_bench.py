#!/usr/bin/env python3
"""Time rml_dnn_trunk alone (csrc/dnn.hip) on resident planes: python tools/trunk_bench.py [--frames 8192] [--hw 80]"""
import argparse
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8192)
    ap.add_argument("--hw", type=int, default=80)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    import torch
    import radar_ml_amd  # noqa: F401
    dnn = importlib.import_module("radar_ml_amd.dnn")
    dev = torch.device("cuda", 0)
    model = dnn.Classifier([(a.hw, a.hw, 1)] * 3, 3).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(1)
    flop = 3 * ((a.hw // 2) ** 2 * 64 * 9 * 2 + (a.hw // 4) ** 2 * 32 * 576 * 2)
    for dt in (torch.bfloat16, torch.float32):
        xs = [(torch.rand((a.frames, a.hw, a.hw), device=dev, generator=g) * 2 - 1).to(dt) for _ in range(3)]
        with torch.no_grad():
            for _ in range(3):
                model.features_fused(*xs)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(a.reps):
                model.features_fused(*xs)
            ev[1].record()
            torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / a.reps
        print(json.dumps({"planes": str(dt), "frames": a.frames, "hw": a.hw, "trunk_ms": round(ms, 4),
                          "frames_per_s": round(a.frames / ms * 1e3), "TFLOPs": round(a.frames * flop / ms / 1e9, 1),
                          "GBs": round(a.frames * (3 * a.hw * a.hw * xs[0].element_size() + (a.hw // 4) ** 2 * 96 * 2) / ms / 1e6, 1)}))


if __name__ == "__main__":
    main()
