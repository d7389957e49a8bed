"""CNN chain with a CU partition (session r5q; review r4 item 4a): projection + preprocessing of pass p + 1 on a stream masked to g CUs
of every XCD, trunk + dense tail of pass p on a stream masked to the other 32 - g.  Streams through hipExtStreamCreateWithCUMask
(mask bit i = local CU i / 8 of XCD i % 8: tools/exp/exp_cumask.hip), wrapped as torch ExternalStreams.  No change to the library:
persistent grids keep their 256-CU sizes (g = 16 divides them evenly)."""
import ctypes, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import radar_ml_amd as rml
from radar_ml_amd import _lib
dnn = importlib.import_module("radar_ml_amd.dnn")
nnc = importlib.import_module("radar_ml_amd.nn_common")
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = dnn.define_classifier(device=dev).eval()
hip = _lib.load()        # libradarml_hip.so depends on libamdhip64: dlsym finds the runtime's symbols through it
create = hip.hipExtStreamCreateWithCUMask
create.restype = ctypes.c_int
create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]


def masked(lo, hi):
    """stream on local CUs [lo, hi) of every XCD"""
    words = (ctypes.c_uint32 * 8)()
    for bit in range(256):
        if lo <= bit // 8 < hi:
            words[bit // 32] |= 1 << (bit % 32)
    h = ctypes.c_void_p()
    rc = create(ctypes.byref(h), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value, device=dev)


B, bs = 65536, 16384
V, _ = rml.synth_volumes(B, 22, 31, 176, seed=5, device=dev)
V8 = V.to(torch.uint8)


def timed(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


def serial(vol):
    return m.predict_volumes(vol, batch_size=bs, label_guard=None)


def partitioned(vol, sA, sB):
    nb = B // bs
    out = torch.empty((B, 3), device=dev)
    cur = torch.cuda.current_stream()
    sA.wait_stream(cur); sB.wait_stream(cur)
    ev_pre = [torch.cuda.Event() for _ in range(2)]
    ev_free = [torch.cuda.Event() for _ in range(2)]
    xs = [None, None]
    with torch.no_grad():
        for b in range(nb + 1):
            if b < nb:
                with torch.cuda.stream(sA):
                    if b >= 2:
                        sA.wait_event(ev_free[b % 2])
                    xs[b % 2] = nnc.preprocess_volumes(vol[b * bs:(b + 1) * bs], (80, 80), mode="max")
                    ev_pre[b % 2].record(sA)
            if b >= 1:
                k = (b - 1) % 2
                with torch.cuda.stream(sB):
                    sB.wait_event(ev_pre[k])
                    out[(b - 1) * bs:b * bs] = m._forward_timed(xs[k], None)
                    ev_free[k].record(sB)
    cur.wait_stream(sA); cur.wait_stream(sB)
    return out


for name, vol in (("f32", V), ("u8", V8)):
    ms0, ref = timed(lambda: serial(vol))
    print("%s serial chain: %.2f ms per %d frames = %.2f M frames/s" % (name, ms0, B, B / ms0 / 1e3))
    plain = [torch.cuda.Stream(), torch.cuda.Stream()]
    ms, out = timed(lambda: partitioned(vol, plain[0], plain[1]))
    print("%s two streams, no mask: %.2f ms (%.2f M)  same bits %s" % (name, ms, B / ms / 1e3, bool(torch.equal(out, ref))))
    for g in (8, 12, 16, 20):
        sA, sB = masked(0, g), masked(g, 32)
        ms, out = timed(lambda: partitioned(vol, sA, sB))
        print("%s partition %2d | %2d CUs per XCD: %.2f ms (%.2f M frames/s)  same bits %s" % (name, g, 32 - g, ms, B / ms / 1e3, bool(torch.equal(out, ref))))
