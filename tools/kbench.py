#!/usr/bin/env python3
"""Kernel-level micro-benchmarks (BASELINE configs[1] projection-only roofline check, and the SVM GEMM).

    python tools/kbench.py proj [--grid 64x64x128] [--frames 4096] [--iters 20]
    python tools/kbench.py svm  [--grid 64x64x128] [--frames 8192] [--svs 2048] [--path i8|f32]
    python tools/kbench.py slice  [--grid ...] [--frames ...] [--u8]     mode SLICE with (i,j,k) given (k_slice_rows)
    python tools/kbench.py derive [--grid ...] [--frames ...] [--u8]     derive only / derive -> slice in one pass (k_derive_slice)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(torch, fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = np.array(ts)
    return float(np.median(ts)), float(ts.min()), float(ts.mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["proj", "svm", "copy", "gemm", "slice", "derive"])
    ap.add_argument("--grid", default="64x64x128")
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--svs", type=int, default=2048)
    ap.add_argument("--path", default="i8")
    ap.add_argument("--mode", default="max")
    ap.add_argument("--u8", action="store_true", help="uint8 volumes (1 byte per voxel)")
    a = ap.parse_args()
    import torch
    import radar_ml_amd as rml
    from radar_ml_amd import _lib
    X, Y, Z = (int(t) for t in a.grid.split("x"))
    D = rml.feature_len(X, Y, Z)
    B = a.frames
    dev = torch.device("cuda", 0)
    if a.what == "copy":
        n = B * X * Y * Z
        src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        med, mn, _ = timeit(torch, lambda: dst.copy_(src), a.iters)
        print(json.dumps({"what": "torch copy", "bytes": 8 * n, "ms": med, "GBs": 8 * n / med / 1e6}))
        med, mn, _ = timeit(torch, lambda: src.amax(), a.iters)
        print(json.dumps({"what": "torch amax (read only)", "bytes": 4 * n, "ms": med, "GBs": 4 * n / med / 1e6}))
        return
    V, cls = rml.synth_volumes(B, X, Y, Z, seed=1)
    esz = 4
    if a.u8:
        V = V.to(torch.uint8)
        esz = 1
    if a.what in ("slice", "derive"):
        lib = _lib.load(); ctx = _lib.context(dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        vdt = 1 if a.u8 else 0
        qb = (D + 127) // 128 * 128
        feat = torch.empty((B, D), dtype=torch.float32, device=dev)
        q = torch.empty((B, qb), dtype=torch.uint8, device=dev)
        isum = torch.empty(B, dtype=torch.int32, device=dev); isq = torch.empty(B, dtype=torch.int64, device=dev)
        flags = torch.empty(B, dtype=torch.int32, device=dev)
        ijk = torch.empty((B, 3), dtype=torch.int32, device=dev)
        _lib.check(lib.rml_derive_targets(ctx, V.data_ptr(), vdt, B, X, Y, Z, 1, ijk.data_ptr(), None, st))
        frame = esz * X * Y * Z
        line = 128                                      # bytes per memory request: every xy value has its own (rows are >= 512 B apart)
        floor_rd = esz * (X * Z + Y * Z) + X * Y * line   # what a slice must fetch whatever the kernel does
        rows = []
        if a.what == "slice":
            def f_rows():
                _lib.check(lib.rml_project(ctx, V.data_ptr(), vdt, B, X, Y, Z, 1, ijk.data_ptr(), 255.0, 7, feat.data_ptr(), D, None, 0,
                                           None, None, None, st))
            def f_codes():
                _lib.check(lib.rml_project(ctx, V.data_ptr(), vdt, B, X, Y, Z, 1, ijk.data_ptr(), 255.0, 7, None, 0, q.data_ptr(), qb,
                                           isum.data_ptr(), isq.data_ptr(), flags.data_ptr(), st))
            for knob in ("1", "0"):
                _lib.set_option("slice_wave", int(knob))
                for label, fn, alg in (("f32 rows /255", f_rows, esz * D + 4 * D + 12), ("codes+stats", f_codes, esz * D + D + 16 + 12)):
                    med, mn, _ = timeit(torch, fn, a.iters)
                    rows.append({"what": "slice (ijk given) %s, %s" % (label, "k_slice_rows" if knob == "1" else "k_project_slice (round 1)"),
                                 "grid": [X, Y, Z], "B": B, "ms_med": round(med, 4), "rows_per_s": round(B / med * 1e3),
                                 "alg_bytes_per_row": alg, "alg_GBs": round(B * alg / med / 1e6, 1),
                                 "request_floor_bytes_per_row": floor_rd + (alg - esz * D), "floor_GBs": round(B * (floor_rd + alg - esz * D) / med / 1e6, 1)})
            _lib.set_option("slice_wave", 1)
        else:
            def f_derive():
                _lib.check(lib.rml_derive_targets(ctx, V.data_ptr(), vdt, B, X, Y, Z, 1, ijk.data_ptr(), None, st))
            def f_fused_codes():
                _lib.check(lib.rml_derive_slice(ctx, V.data_ptr(), vdt, B, X, Y, Z, 1, ijk.data_ptr(), None, 255.0, 7, None, 0, q.data_ptr(), qb,
                                                isum.data_ptr(), isq.data_ptr(), flags.data_ptr(), st))
            def f_fused_rows():
                _lib.check(lib.rml_derive_slice(ctx, V.data_ptr(), vdt, B, X, Y, Z, 1, ijk.data_ptr(), None, 255.0, 7, feat.data_ptr(), D, None, 0,
                                                None, None, None, st))
            for knob in ("1", "0"):
                _lib.set_option("derive_fused", int(knob))
                _lib.set_option("slice_wave", int(knob))
                for label, fn, alg in (("derive only", f_derive, frame + 12), ("derive -> slice, codes+stats", f_fused_codes, frame + 16),
                                       ("derive -> slice, f32 rows", f_fused_rows, frame + 4 * D + 12)):
                    med, mn, _ = timeit(torch, fn, a.iters)
                    rows.append({"what": "%s (%s)" % (label, "k_derive_slice" if knob == "1" else "sum planes + k_profiles_topk + k_project_slice"),
                                 "grid": [X, Y, Z], "B": B, "ms_med": round(med, 4), "ms_min": round(mn, 4), "frames_per_s": round(B / med * 1e3),
                                 "alg_GBs": round(B * alg / med / 1e6, 1), "frac_of_8TBs": round(B * alg / med / 1e6 / 8000, 4)})
            _lib.set_option("derive_fused", 1); _lib.set_option("slice_wave", 1)
        for r in rows:
            print(json.dumps(r))
        return
    if a.what == "gemm":
        # the exact-integer GEMM + finish alone, on code rows (the operand the fused pipeline hands it)
        M = a.svs
        Bm = max(B, M)
        Vm, _ = rml.synth_volumes(Bm, X, Y, Z, seed=1)
        _, q, isum, isq, flags = rml.process_volumes(Vm, mode="max", scale=True, codes=True)
        del Vm
        svq = (q[:M, :D] ^ 0x80).cpu().numpy()
        sv = (svq.astype(np.float32) / np.float32(255.0)).astype(np.float64)
        ns = np.array([M // 3, M // 3, M - 2 * (M // 3)], dtype=np.int32)
        rng = np.random.default_rng(0)
        svc = rml.GpuSVC(sv, rng.uniform(-10, 10, (2, M)), np.array([0.1, -0.2, 0.3]), ns, 0.01, np.arange(3), calib_a=-np.ones(3), calib_b=np.zeros(3))
        ld = svc_ld = None
        # the model's code-row stride (odd multiple of 128 B)
        kq = (D + 127) // 128 * 128
        ldq = kq if (kq // 128) % 2 else kq + 128
        qq = torch.zeros((B, ldq), dtype=torch.uint8, device=dev)
        qq[:, :q.shape[1]] = q[:B]
        qq[:, D:] = 0
        fn = lambda: svc.decide_codes(qq, isum[:B], isq[:B], flags[:B], want_proba=True)
        med, mn, mean = timeit(torch, fn, a.iters)
        ops = 2.0 * B * M * D
        print(json.dumps({"what": "exact GEMM + finish on code rows", "N": B, "M": M, "D": D, "ms_med": round(med, 4), "ms_min": round(mn, 4),
                          "T_op_s": round(ops / med / 1e9, 1), "frac_of_3944": round(ops / med / 1e9 / 3944, 4)}))
        return
    if a.what == "proj":
        feat = torch.empty((B, D), dtype=torch.float32, device=dev)
        res = []
        qb = (D + 127) // 128 * 128
        q = torch.empty((B, qb), dtype=torch.uint8, device=dev)
        for label, kw, outbytes in [("f32 rows", dict(out=feat), 4 * D), ("f32 rows /255", dict(out=feat, scale=True), 4 * D),
                                    ("f32 rows + codes", dict(out=feat, codes=True), 4 * D + (D + 127) // 128 * 128)]:
            med, mn, mean = timeit(torch, lambda: rml.process_volumes(V, mode=a.mode, **kw), a.iters)
            alg = B * (esz * X * Y * Z + outbytes)
            res.append({"what": "project %s %s" % (a.mode, label), "grid": [X, Y, Z], "B": B, "ms_med": round(med, 4),
                        "ms_min": round(mn, 4), "frames_per_s": round(B / med * 1e3), "alg_GBs": round(alg / med / 1e6, 1),
                        "frac_of_8TBs": round(alg / med / 1e6 / 8000, 4)})
        # the fused pipeline's first pass: codes + row statistics only (no float rows), through the C ABI
        lib = _lib.load(); ctx = _lib.context(dev)
        isum = torch.empty(B, dtype=torch.int32, device=dev); isq = torch.empty(B, dtype=torch.int64, device=dev)
        flags = torch.empty(B, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        def codes_only():
            _lib.check(lib.rml_project(ctx, V.data_ptr(), 1 if a.u8 else 0, B, X, Y, Z, 0, None, 255.0, 7, None, 0, q.data_ptr(), qb,
                                       isum.data_ptr(), isq.data_ptr(), flags.data_ptr(), st))
        med, mn, mean = timeit(torch, codes_only, a.iters)
        alg = B * (esz * X * Y * Z + 16)
        res.append({"what": "project max codes+stats only", "grid": [X, Y, Z], "B": B, "ms_med": round(med, 4), "ms_min": round(mn, 4),
                    "frames_per_s": round(B / med * 1e3), "alg_GBs": round(alg / med / 1e6, 1), "frac_of_8TBs": round(alg / med / 1e6 / 8000, 4)})
        for r in res:
            print(json.dumps(r))
    else:
        rng = np.random.default_rng(0)
        M = a.svs
        feat, q, isum, isq, flags = rml.process_volumes(V, mode="max", scale=True, codes=True)
        svq = (q[:M, :D] ^ 0x80).cpu().numpy() if M <= B else None
        sv = (svq.astype(np.float32) / np.float32(255.0)).astype(np.float64)
        ns = np.array([M // 3, M // 3, M - 2 * (M // 3)], dtype=np.int32)
        dc = rng.uniform(-10, 10, (2, M))
        svc = rml.GpuSVC(sv, dc, np.array([0.1, -0.2, 0.3]), ns, 0.01, np.arange(3), path=a.path)
        Xd = feat
        med, mn, mean = timeit(torch, lambda: svc._decide(Xd), max(3, a.iters // 4))
        ops = 2.0 * B * M * D
        print(json.dumps({"what": "svm decision path=%s (incl. row prep)" % a.path, "N": B, "M": M, "D": D, "ms_med": round(med, 3),
                          "frames_per_s": round(B / med * 1e3), "T_op_s": round(ops / med / 1e9, 1)}))


if __name__ == "__main__":
    main()
