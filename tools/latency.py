#!/usr/bin/env python3
"""Single-observation latency of the drop-in surface (the way predict.py:98-119 calls it: one target per call).

    python tools/latency.py [--grid 22x31x176] [--svs 2000]
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
    ap.add_argument("--svs", type=int, default=2000)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    import torch
    import radar_ml_amd as rml
    X, Y, Z = (int(t) for t in a.grid.split("x"))
    D = rml.feature_len(X, Y, Z)
    rng = np.random.default_rng(0)
    M = a.svs
    V, _ = rml.synth_volumes(max(M, 256), X, Y, Z, seed=3)
    feat, q, *_ = rml.process_volumes(V, mode="max", scale=True, codes=True)
    sv = ((q[:M, :D] ^ 0x80).cpu().numpy().astype(np.float32) / np.float32(255.0)).astype(np.float64)
    ns = np.array([M // 3, M // 3, M - 2 * (M // 3)], dtype=np.int32)
    svc = rml.GpuSVC(sv, rng.uniform(-1, 1, (2, M)), np.array([0.1, -0.2, 0.3]), ns, 0.01, np.arange(3),
                     calib_a=np.array([-1.0, -1.1, -0.9]), calib_b=np.array([0.0, 0.1, -0.1]))

    def timed(fn, sync=True):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.iters):
            t0 = time.perf_counter()
            fn()
            if sync:
                torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e6
        return {"p50_us": round(float(np.percentile(ts, 50)), 1), "p99_us": round(float(np.percentile(ts, 99)), 1)}

    cal = rml.GpuCalibratedClassifier(svc)
    v1 = V[:1].contiguous()
    v1_u8 = v1.to(torch.uint8)
    v1_host = v1.cpu().numpy()
    ijk1 = torch.tensor([[X // 2, Y // 2, Z // 3]], dtype=torch.int32, device=v1.device)
    row_host = feat[:1].cpu().numpy()
    res = {
        "grid": [X, Y, Z], "n_sv": M,
        "volume_on_gpu_to_labels (decide_volumes, B=1)": timed(lambda: svc.decide_volumes(v1)),
        "uint8_volume_on_gpu_to_labels (B=1)": timed(lambda: svc.decide_volumes(v1_u8)),
        "slice_at_sdk_target_to_labels (predict.py:98-119, B=1)": timed(lambda: svc.decide_volumes(v1, mode="slice", ijk=ijk1, validate_ijk=False)),
        "slice_at_derived_target_to_labels (B=1)": timed(lambda: svc.decide_volumes(v1, mode="slice")),
        "host_volume_to_host_proba (B=1, incl. PCIe)": timed(lambda: svc.decide_volumes(v1_host)["proba"].cpu()),
        "host_feature_row_predict_proba (predict.py:60)": timed(lambda: cal.predict_proba(row_host)),
        "batch_64_volumes_on_gpu": timed(lambda: svc.decide_volumes(V[:64])),
    }
    print(json.dumps(res))


if __name__ == "__main__":
    main()
