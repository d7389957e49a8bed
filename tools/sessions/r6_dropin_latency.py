#!/usr/bin/env python3
"""The per-target body of predict.py:98-119 through the drop-in surface, host clock per target:
slices of a HOST raw image -> rml.process_samples -> rml.classifier (GpuCalibratedClassifier.predict_proba on a host row)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import radar_ml_amd as rml

def main():
    for grid, M in (((22, 31, 176), 2000), ((64, 64, 128), 2560)):
        X, Y, Z = grid
        D = rml.feature_len(X, Y, Z)
        rng = np.random.default_rng(0)
        V, _ = rml.synth_volumes(max(M, 256), X, Y, Z, seed=3)
        _, q, *_ = rml.process_volumes(V, mode="max", scale=True, codes=True)
        sv = ((q[:M, :D] ^ 0x80).cpu().numpy().astype(np.float32) / np.float32(255.0)).astype(np.float64)
        ns = np.array([M // 3, M // 3, M - 2 * (M // 3)], dtype=np.int32)
        svc = rml.GpuSVC(sv, rng.uniform(-1, 1, (2, M)), np.array([0.1, -0.2, 0.3]), ns, 0.01, np.arange(3),
                         calib_a=np.array([-1.0, -1.1, -0.9]), calib_b=np.array([0.0, 0.1, -0.1]))
        model = rml.GpuCalibratedClassifier(svc)
        class LE: classes_ = np.array(["a", "b", "c"])
        raw = V[0].cpu().numpy()
        i, j, k = X // 2, Y // 2, Z // 3
        mask = rml.ProjMask(True, True, True)
        zoom = rml.calc_proj_zoom(X, Y, Z, X, Y, Z) if hasattr(rml, "calc_proj_zoom") else None

        def body():
            yz, xz, xy = raw[i, :, :], raw[:, j, :], raw[:, :, k]
            obs = rml.process_samples([(xz, yz, xy)], proj_mask=mask, scale=True) if zoom is None else \
                rml.process_samples([(xz, yz, xy)], proj_mask=mask, proj_zoom=zoom, scale=True)
            return rml.classifier(obs, model, LE, 0.7)

        def part(fn, n=200):
            for _ in range(10): fn()
            ts = []
            for _ in range(n):
                t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
            return round(float(np.percentile(np.array(ts) * 1e6, 50)), 1)
        obs = rml.process_samples([(raw[:, j, :], raw[i, :, :], raw[:, :, k])], proj_mask=mask, scale=True)
        print(grid, {"per_target_body_us": part(body),
                     "process_samples_us": part(lambda: rml.process_samples([(raw[:, j, :], raw[i, :, :], raw[:, :, k])], proj_mask=mask, scale=True)),
                     "classifier_us": part(lambda: rml.classifier(obs, model, LE, 0.7))})

if __name__ == "__main__":
    main()
