#!/usr/bin/env python3
"""Same-process A/B of the SVM GEMM kernels (one box, interleaved rounds; cdna_hip_programming.md rule 24).

    python tools/gemm_ab.py exact  --grid 64x64x128 --svs 2562 --frames 16384,17664 [--rounds 5]
    python tools/gemm_ab.py digits --grid 64x64x128 --svs 2562 --frames 16384 [--rounds 3]

exact : code rows -> k_svm_gemm_ring (5-slot operand-stage ring, interleaved DMA issue) vs k_svm_gemm_i8_256 (two 64 KiB
        stages) vs the 128 x 128 kernel, each as ONE launch over the whole batch (RML_OPT_CHUNK pinned to the batch), GEMM + finish.
digits: float rows off the code grid -> RML_PATH_DIGITS (ten int8 digit-plane products) vs RML_PATH_F64 (float64 MFMA),
        row preparation included, plus the largest |dec| difference between the two.
RML_LIB selects a variant build (e.g. the library-exp() build for the epilogue A/B).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(torch, fn, iters):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["exact", "digits"])
    ap.add_argument("--grid", default="64x64x128")
    ap.add_argument("--frames", default="16384")
    ap.add_argument("--svs", type=int, default=2562)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    import torch
    import radar_ml_amd as rml
    from radar_ml_amd import _lib
    X, Y, Z = (int(t) for t in a.grid.split("x"))
    D = rml.feature_len(X, Y, Z)
    M = a.svs
    dev = torch.device("cuda", 0)
    frames = [int(t) for t in a.frames.split(",")]
    Bmax = max(max(frames), M)
    rng = np.random.default_rng(0)
    V, _ = rml.synth_volumes(Bmax, X, Y, Z, seed=1)
    feat, q, isum, isq, flags = rml.process_volumes(V, mode="max", scale=True, codes=True)
    del V
    svq = (q[:M, :D] ^ 0x80).cpu().numpy()
    sv = (svq.astype(np.float32) / np.float32(255.0)).astype(np.float64)
    ns = np.array([M // 3, M // 3, M - 2 * (M // 3)], dtype=np.int32)
    dc = rng.uniform(-10, 10, (2, M))
    ic = np.array([0.1, -0.2, 0.3])
    lib = os.environ.get("RML_LIB", "default")
    if a.what == "exact":
        svc = rml.GpuSVC(sv, dc, ic, ns, 0.01, np.arange(3), calib_a=-np.ones(3), calib_b=np.zeros(3))
        kq = (D + 127) // 128 * 128
        ldq = kq if (kq // 128) % 2 else kq + 128
        for B in frames:
            qq = torch.zeros((B, ldq), dtype=torch.uint8, device=dev)
            qq[:, :q.shape[1]] = q[:B]
            qq[:, D:] = 0
            _lib.set_option("chunk", (B + 127) // 128 * 128)
            fn = lambda: svc.decide_codes(qq, isum[:B], isq[:B], flags[:B], want_proba=True)
            arms = {"ring": {"gemm_big": 1}, "tile128": {"gemm_big": 0}}       # (the two-stage 256 x 256 arm left the tree in round 4)
            res = {k: [] for k in arms}
            outs = {}
            for r in range(a.rounds):
                for name, env in arms.items():
                    for k_, v_ in env.items():
                        _lib.set_option(k_, v_)
                    med, mn = timed(torch, fn, a.iters)
                    res[name].append(med)
                    if r == 0:
                        outs[name] = fn()[0].cpu().numpy()
            ops = 2.0 * B * M * D
            tiles = ((B + 255) // 256) * ((M + 255) // 256)
            row = {"what": "exact", "lib": lib, "grid": [X, Y, Z], "N": B, "M": M, "D": D, "tiles256": tiles, "rounds_of_256cu": round(tiles / 256, 3)}
            for name in arms:
                m = float(np.median(res[name]))
                row[name] = {"ms": round(m, 4), "ms_min": round(float(np.min(res[name])), 4), "POPs": round(ops / m / 1e12, 3),
                             "frac_of_3944": round(ops / m / 1e9 / 3944, 4)}
            row["max_abs_diff_128_vs_256"] = float(np.abs(outs["ring"] - outs["tile128"]).max())
            print(json.dumps(row), flush=True)
            del qq
    else:
        off = np.float64(0.9990234375)
        svg = sv * off
        svc = rml.GpuSVC(svg, dc, ic, ns, 0.01, np.arange(3), calib_a=-np.ones(3), calib_b=np.zeros(3))
        assert not svc.exact
        for B in frames:
            Xd = (feat[:B] * float(off)).contiguous()
            noise = torch.randn(Xd.shape, device=dev, dtype=torch.float32) * 1e-3
            Xd = Xd + noise * (Xd > 0)
            del noise
            _lib.set_option("chunk", (B + 127) // 128 * 128)
            arms = {"digits": "digits", "f64": "f64"}
            res = {k: [] for k in arms}
            outs = {}
            for r in range(a.rounds):
                for name, path in arms.items():
                    med, mn = timed(torch, lambda: svc._decide(Xd, want_proba=True, path=path), max(2, a.iters // 2))
                    res[name].append(med)
                    if r == 0:
                        outs[name] = svc._decide(Xd, path=path)[0].cpu().numpy()
            ops = 2.0 * B * M * D
            row = {"what": "digits", "lib": lib, "grid": [X, Y, Z], "N": B, "M": M, "D": D}
            for name in arms:
                m = float(np.median(res[name]))
                row[name] = {"ms": round(m, 3), "frames_per_s": round(B / m * 1e3), "eff_Tflop_s": round(ops / m / 1e9, 1)}
            row["digits_int8_POPs"] = round(10 * ops / float(np.median(res["digits"])) / 1e12, 3)
            row["max_abs_dec_diff"] = float(np.abs(outs["digits"] - outs["f64"]).max())
            print(json.dumps(row), flush=True)
            del Xd


if __name__ == "__main__":
    main()
