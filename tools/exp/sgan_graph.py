#!/usr/bin/env python3
"""Experiment: capture the SGAN discriminator fwd+bwd (fp16 autocast) in HIP graphs (torch.cuda.graphs), optimizer
(fused Adam) + GradScaler outside; compare with the eager step."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import radar_ml_amd as rml  # noqa

sgan = importlib.import_module("radar_ml_amd.sgan")
dev = torch.device("cuda", 0)
n = 256
torch.manual_seed(0)
d = sgan.define_discriminator(device=dev)
g = torch.Generator(device=dev).manual_seed(0)
x = [(torch.rand((n, 1, 128, 128), device=dev, generator=g) * 2 - 1).contiguous(memory_format=torch.channels_last) for _ in range(3)]
y = torch.randint(0, 3, (n,), device=dev, generator=g)
yr = torch.full((n,), 0.9, device=dev)
opt_c = torch.optim.Adam(d.parameters(), lr=2e-4, betas=(0.5, 0.999), eps=1e-7, fused=True)
opt_d = torch.optim.Adam(d.parameters(), lr=2e-4, betas=(0.5, 0.999), eps=1e-7, fused=True)
scaler = torch.amp.GradScaler("cuda")
d.train()


def fwd_bwd(loss_fn):
    with torch.autocast("cuda", dtype=torch.float16, cache_enabled=False):
        lg = d(*x)
    loss = loss_fn(lg)
    scaler.scale(loss).backward()
    return loss


lc_fn = lambda lg: sgan.c_loss(lg, y)
ld_fn = lambda lg: sgan.d_loss(lg, yr)

# warm-up on a side stream (MIOpen find, allocator)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        for opt, fn in ((opt_c, lc_fn), (opt_d, ld_fn)):
            opt.zero_grad(set_to_none=False)
            fwd_bwd(fn)
            scaler.step(opt); scaler.update()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()


def eager_step():
    for opt, fn in ((opt_c, lc_fn), (opt_d, ld_fn)):
        opt.zero_grad(set_to_none=False)
        fwd_bwd(fn)
        scaler.step(opt); scaler.update()


def timeit(fn, k=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


print("eager (fused Adam, zero_grad in place): %.2f ms/step" % timeit(eager_step), flush=True)

graphs = {}
for name, opt, fn in (("c", opt_c, lc_fn), ("d", opt_d, ld_fn)):
    gph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=False)
    with torch.cuda.graph(gph):
        loss = fwd_bwd(fn)
    graphs[name] = (gph, loss, opt)


def graph_step():
    for name in ("c", "d"):
        gph, loss, opt = graphs[name]
        opt.zero_grad(set_to_none=False)
        gph.replay()
        scaler.step(opt); scaler.update()


print("graphed fwd+bwd: %.2f ms/step" % timeit(graph_step), flush=True)
print("losses", float(graphs["c"][1]), float(graphs["d"][1]))
