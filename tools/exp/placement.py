#!/usr/bin/env python3
"""Does the rate of the streaming projection depend on WHERE the volumes live?  One process, the same kernel (codes + statistics, Walabot grid
or 64x64x128 in the pipeline's configuration), the volumes re-allocated several times -- fresh allocations, allocations at byte offsets into one
big block -- and the rate of each placement.  python tools/exp/placement.py [--grid 22x31x176] [--frames 16384]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="22x31x176"); ap.add_argument("--frames", type=int, default=16384)
a = ap.parse_args()
import torch
import radar_ml_amd as rml
from radar_ml_amd import _lib
X, Y, Z = (int(t) for t in a.grid.split("x"))
B = a.frames
D = rml.feature_len(X, Y, Z)
dev = torch.device("cuda", 0)
lib = _lib.load(); ctx = _lib.context(dev)
st = torch.cuda.current_stream(dev).cuda_stream
qb = (D + 127) // 128 * 128
q = torch.empty((B, qb), dtype=torch.uint8, device=dev)
isum = torch.empty(B, dtype=torch.int32, device=dev); isq = torch.empty(B, dtype=torch.int64, device=dev); flags = torch.empty(B, dtype=torch.int32, device=dev)
n = B * X * Y * Z
src, _ = rml.synth_volumes(B, X, Y, Z, seed=1)
src = src.reshape(-1)

def rate(v, qq=None):
    qq = q if qq is None else qq
    def f():
        _lib.check(lib.rml_project(ctx, v.data_ptr(), 0, B, X, Y, Z, 0, None, 255.0, 7, None, 0, qq.data_ptr(), qb, isum.data_ptr(), isq.data_ptr(), flags.data_ptr(), st))
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); ms = ts[len(ts) // 2]
    return B * (4 * X * Y * Z + 16) / ms / 1e6 / 8000

print("grid %s, %d frames (%.2f GB)" % (a.grid, B, n * 4 / 1e9))
print("the generator's own tensor           ptr %% 2MiB = %8d   frac %.4f" % (src.data_ptr() % (1 << 21), rate(src)))
keep = []
for i in range(4):
    v = torch.empty(n, dtype=torch.float32, device=dev); v.copy_(src)
    print("fresh allocation %d                   ptr %% 2MiB = %8d   frac %.4f" % (i, v.data_ptr() % (1 << 21), rate(v)))
    keep.append(v)          # keep them: the next one lands elsewhere
del keep
torch.cuda.empty_cache()
big = torch.empty(n + (64 << 20), dtype=torch.float32, device=dev)
for off in (0, 64, 1024, 4096, 65536, 1 << 20, (1 << 21) // 4 * 1, 3 << 20, 16 << 20):        # offsets in floats
    v = big[off:off + n]; v.copy_(src)
    print("one block, offset %9d B         ptr %% 2MiB = %8d   frac %.4f" % (off * 4, v.data_ptr() % (1 << 21), rate(v)))

# the other way round: the volumes stay, the code rows move
del big
torch.cuda.empty_cache()
v = torch.empty(n, dtype=torch.float32, device=dev); v.copy_(src)
keep = []
for i in range(6):
    qq = torch.empty((B, qb), dtype=torch.uint8, device=dev)
    print("volumes fixed, code rows allocation %d (%.0f MB apart)   frac %.4f" % (i, B * qb / 1e6, rate(v, qq)))
    keep.append(qq)
    keep.append(torch.empty(((i + 1) * 300 << 20,), dtype=torch.uint8, device=dev))      # spacers of growing size
