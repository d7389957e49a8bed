#!/usr/bin/env python3
"""Measuring stick for k_svm_gemm: the library int8 GEMM (torch._int_mm -> hipBLASLt) at the same shape
(8192 samples x 2560 SVs x 20480 codes), no epilogue."""
import time
import torch

N, M, D = 8192, 2560, 20480
a = torch.randint(-128, 127, (N, D), dtype=torch.int8, device="cuda")
b = torch.randint(-128, 127, (M, D), dtype=torch.int8, device="cuda")
bt = b.t()
for name, fn in (("_int_mm(a, b.T)", lambda: torch._int_mm(a, bt)),
                 ("_int_mm(a, b.T.contiguous())", None)):
    try:
        if fn is None:
            btc = b.t().contiguous()
            fn = lambda: torch._int_mm(a, btc)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("%s: %.3f ms, %.2f PetaOP/s" % (name, dt * 1e3, 2.0 * N * M * D / dt / 1e15))
    except Exception as e:
        print(name, "failed:", str(e)[:200])
