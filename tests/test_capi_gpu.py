"""C-ABI behaviour on the GPU: status codes and messages for bad arguments, context lifecycle, streams."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_error_codes_and_messages(rml):
    import torch
    from radar_ml_amd import _lib
    lib = _lib.load()
    ctx = _lib.context()
    v = torch.zeros((2, 4, 4, 8), device="cuda")
    feat = torch.empty((2, 4 * 8 + 4 * 8 + 16), device="cuda")
    st = _lib.stream_ptr()
    # empty mask, NULL volume, ld too small, slice without ijk, unknown mode
    assert lib.rml_project(ctx, _lib.ptr(v), 0, 2, 4, 4, 8, 0, None, 0.0, 0, _lib.ptr(feat), feat.stride(0), None, 0, None, None, None, st) == -1
    assert b"mask" in lib.rml_last_error()
    assert lib.rml_project(ctx, None, 0, 2, 4, 4, 8, 0, None, 0.0, 7, _lib.ptr(feat), feat.stride(0), None, 0, None, None, None, st) == -1
    assert lib.rml_project(ctx, _lib.ptr(v), 0, 2, 4, 4, 8, 0, None, 0.0, 7, _lib.ptr(feat), 5, None, 0, None, None, None, st) == -1
    assert b"ld_feat" in lib.rml_last_error()
    assert lib.rml_project(ctx, _lib.ptr(v), 0, 2, 4, 4, 8, 1, None, 0.0, 7, _lib.ptr(feat), feat.stride(0), None, 0, None, None, None, st) == -1
    assert lib.rml_project(ctx, _lib.ptr(v), 0, 2, 4, 4, 8, 9, None, 0.0, 7, _lib.ptr(feat), feat.stride(0), None, 0, None, None, None, st) == -1
    assert lib.rml_project(ctx, _lib.ptr(v), 5, 2, 4, 4, 8, 0, None, 0.0, 7, _lib.ptr(feat), feat.stride(0), None, 0, None, None, None, st) == -1
    assert b"dtype" in lib.rml_last_error()
    # B == 0 is a no-op even with NULL pointers
    assert lib.rml_project(ctx, None, 0, 0, 4, 4, 8, 0, None, 0.0, 7, None, 0, None, 0, None, None, None, st) == 0
    # model load validation
    sv = np.zeros((4, 8)); dc = np.zeros((2, 4)); ic = np.zeros(3); h = C.c_void_p()
    ns_bad = np.array([1, 1, 1], dtype=np.int32)
    assert lib.rml_svm_load(ctx, sv.ctypes.data, 4, 8, dc.ctypes.data, ic.ctypes.data, ns_bad.ctypes.data, 3, 0, 0.1, 255.0, None, None, C.byref(h)) == -1
    assert b"n_support" in lib.rml_last_error() and not h.value
    ns = np.array([2, 1, 1], dtype=np.int32)
    assert lib.rml_svm_load(ctx, sv.ctypes.data, 4, 8, dc.ctypes.data, ic.ctypes.data, ns.ctypes.data, 9, 0, 0.1, 255.0, None, None, C.byref(h)) == -2
    assert lib.rml_svm_load(ctx, sv.ctypes.data, 4, 8, dc.ctypes.data, ic.ctypes.data, ns.ctypes.data, 3, 7, 0.1, 255.0, None, None, C.byref(h)) == -2
    assert lib.rml_svm_load(ctx, sv.ctypes.data, 4, 8, dc.ctypes.data, ic.ctypes.data, ns.ctypes.data, 3, 0, 0.1, 255.0, None, None, C.byref(h)) == 0
    assert lib.rml_svm_num_sv(h) == 4 and lib.rml_svm_dim(h) == 8 and lib.rml_svm_is_exact(h) == 1
    x = torch.zeros((3, 8), device="cuda")
    dec = torch.empty((3, 3), dtype=torch.float64, device="cuda")
    pr = torch.empty((3, 3), dtype=torch.float64, device="cuda")
    # proba without calibrators -> state error; wrong D for the fused door -> invalid
    assert lib.rml_svm_decision(ctx, h, 0, _lib.ptr(x), 8, None, 0, None, None, None, 3, _lib.ptr(dec), None, _lib.ptr(pr), None, None, st) == -5
    assert lib.rml_project_svm(ctx, h, _lib.ptr(v), 0, 2, 4, 4, 8, 0, None, 255.0, 7, _lib.ptr(dec), None, None, None, None, st) == -1
    assert b"D=" in lib.rml_last_error()
    assert lib.rml_svm_decision(ctx, h, 0, _lib.ptr(x), 8, None, 0, None, None, None, 3, _lib.ptr(dec), None, None, None, None, st) == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dec.cpu().numpy(), 0.0)          # all-zero SVs and coefficients, zero intercepts
    assert lib.rml_svm_free(ctx, h) == 0


def test_context_lifecycle_and_side_stream(rml):
    import torch
    from radar_ml_amd import _lib
    import oracle_np as O
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.rml_ctx_create(99, C.byref(h)) == -1 and not h.value
    assert lib.rml_ctx_create(0, C.byref(h)) == 0 and lib.rml_ctx_device(h) == 0
    assert lib.rml_ctx_destroy(h) == 0
    assert lib.rml_ctx_destroy(None) == 0
    # launches honour the caller's stream
    v, _ = O.synth_volumes(3, 16, 22, 31, 176)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        xz, yz, xy = rml.project(torch.from_numpy(v).cuda(), mode="max")
    s.synchronize()
    for g, w in zip((xz, yz, xy), O.project_max(v)):
        np.testing.assert_array_equal(g.cpu().numpy(), w)


def test_new_entry_points_reject_bad_arguments(rml):
    """rml_resize_bicubic / rml_dnn_trunk / rml_bn_lrelu_pad_*: status codes and messages, no launch on bad input."""
    import torch
    from radar_ml_amd import _lib
    lib = _lib.load()
    ctx = _lib.context()
    st = _lib.stream_ptr()
    x = torch.zeros((2, 8, 8), device="cuda")
    out = torch.zeros((2, 4, 4), device="cuda")
    assert lib.rml_resize_bicubic(ctx, None, 64, 2, 8, 8, 4, 4, 0.0, 0.0, _lib.ptr(out), 0, st) == -1
    assert lib.rml_resize_bicubic(ctx, _lib.ptr(x), 10, 2, 8, 8, 4, 4, 0.0, 0.0, _lib.ptr(out), 0, st) == -1     # in_stride < H*W
    assert b"in_stride" in lib.rml_last_error()
    assert lib.rml_resize_bicubic(ctx, _lib.ptr(x), 64, 2, 8, 8, 0, 4, 0.0, 0.0, _lib.ptr(out), 0, st) == -1
    assert lib.rml_resize_bicubic(ctx, None, 64, 0, 8, 8, 4, 4, 0.0, 0.0, None, 0, st) == 0                     # empty batch
    feat = torch.zeros((2, 2 * 2 * 96), dtype=torch.bfloat16, device="cuda")
    w1 = torch.zeros((3, 64, 9), device="cuda"); b1 = torch.zeros((3, 64), device="cuda")
    w2 = torch.zeros((3, 32, 576), dtype=torch.bfloat16, device="cuda"); b2 = torch.zeros((3, 32), device="cuda")
    p6 = torch.zeros((2, 6, 8), device="cuda")
    assert lib.rml_dnn_trunk(ctx, _lib.ptr(p6), _lib.ptr(p6), _lib.ptr(p6), 0, 2, 6, 8, _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2),
                             _lib.ptr(feat), st) == -2                                                            # H not a multiple of 4
    assert b"multiple" in lib.rml_last_error()
    assert lib.rml_dnn_trunk(ctx, None, None, None, 0, 2, 8, 8, _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(feat), st) == -1
    xh = torch.zeros((2, 4, 4, 12), dtype=torch.float16, device="cuda")
    g = torch.ones(12, device="cuda"); z = torch.zeros(12, device="cuda")
    ws = torch.zeros(int(lib.rml_bn_workspace_floats(ctx, 16)) + 64, device="cuda")
    assert lib.rml_bn_lrelu_pad_forward(ctx, _lib.ptr(xh), 0, 2, 4, 4, 12, 1, 1, _lib.ptr(g), _lib.ptr(z), 1e-3, 0.01, 0.2, None, None,
                                        _lib.ptr(z), _lib.ptr(z), _lib.ptr(ws), _lib.ptr(xh), st) == -2               # C = 12
    assert lib.rml_bn_lrelu_pad_forward(ctx, _lib.ptr(xh), 7, 2, 4, 4, 16, 1, 1, _lib.ptr(g), _lib.ptr(z), 1e-3, 0.01, 0.2, None, None,
                                        _lib.ptr(z), _lib.ptr(z), _lib.ptr(ws), _lib.ptr(xh), st) == -1               # dtype


def test_round2_entry_points_reject_bad_arguments(rml):
    """rml_project_slices / rml_svm_set_platt / rml_augment: status codes and messages; and two host threads sharing the
    context (the workspace users serialise on it: include/radarml.h conventions)."""
    import threading
    import torch
    from radar_ml_amd import _lib
    import oracle_np as O
    from conftest import load_golden, svm_model_arrays
    lib = _lib.load()
    ctx = _lib.context()
    st = _lib.stream_ptr()
    v = torch.zeros((2, 4, 4, 8), device="cuda")
    D = 4 * 8 + 4 * 8 + 16
    feat = torch.empty((6, D), device="cuda")
    ijk = torch.zeros((6, 3), dtype=torch.int32, device="cuda")
    assert lib.rml_project_slices(ctx, _lib.ptr(v), 0, 2, 4, 4, 8, 0, _lib.ptr(ijk), 0.0, 7, _lib.ptr(feat), D, None, 0, None, None, None, st) == -1
    assert b"target" in lib.rml_last_error()
    assert lib.rml_project_slices(ctx, _lib.ptr(v), 0, 2, 4, 4, 8, 3, None, 0.0, 7, _lib.ptr(feat), D, None, 0, None, None, None, st) == -1
    assert lib.rml_project_slices(ctx, _lib.ptr(v), 0, 2, 4, 4, 8, 3, _lib.ptr(ijk), 0.0, 7, _lib.ptr(feat), D, None, 0, None, None, None, st) == 0
    p = torch.zeros((2, 8, 8), device="cuda"); q = torch.empty_like(p)
    par = torch.zeros((2, 6), dtype=torch.float64, device="cuda")
    assert lib.rml_augment(ctx, 7, _lib.ptr(p), 2, 8, 8, _lib.ptr(par), _lib.ptr(q), st) == -1 and b"op" in lib.rml_last_error()
    assert lib.rml_augment(ctx, 0, _lib.ptr(p), 2, 8, 8, None, _lib.ptr(q), st) == -1
    assert lib.rml_augment(ctx, 0, _lib.ptr(p), 2, 200, 200, _lib.ptr(par), _lib.ptr(q), st) == -2          # plane larger than the LDS
    assert lib.rml_augment(ctx, 0, None, 0, 8, 8, None, None, st) == 0
    g = load_golden("svm_small.npz")
    svc = rml.GpuSVC(*[svm_model_arrays(g)[k] for k in ("sv", "dual_coef", "intercept", "n_support", "gamma", "classes")])
    assert lib.rml_svm_set_platt(ctx, svc._h, None, None) == -1
    # two threads, two streams, one context: results must be what one thread gets
    m = svm_model_arrays(g)
    X = g["test_feat_u8"].astype(np.float32) / np.float32(255.0)
    want = svc.decision_function(X)
    vol, _ = O.synth_volumes(2, 64, 8, 10, 16)
    wantp = O.project_max(vol)
    errs = []

    def work(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(20):
                    if i == 0:
                        got = svc.decision_function(X)
                        assert np.array_equal(got, want)
                    else:
                        t = rml.derive_targets(torch.from_numpy(vol).cuda(), 2).cpu().numpy()
                        assert t.shape == (64, 2, 3)
                        got = rml.project(vol, mode="max")
                        assert all(np.array_equal(a, b) for a, b in zip(got, wantp))
        except Exception as e:          # pragma: no cover
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errs, errs


def test_round4_entry_points_reject_bad_arguments_and_options(rml):
    """rml_derive_slice / rml_derive_project_svm / rml_derive_slice_supported / rml_ctx_set_option through the C ABI: status
    codes and messages for bad arguments, the support predicate, and the projection option (same results with it on)."""
    import torch
    from radar_ml_amd import _lib
    lib = _lib.load()
    ctx = _lib.context()
    st = _lib.stream_ptr()
    X, Y, Z = 4, 6, 16
    D = X * Z + Y * Z + X * Y
    v = torch.zeros((3, X, Y, Z), device="cuda")
    feat = torch.empty((3, D), device="cuda")
    ijk = torch.empty((3, 3), dtype=torch.int32, device="cuda")
    ok = lambda nt, mask=7, ld=D, vol=v: lib.rml_derive_slice(ctx, _lib.ptr(vol), 0, 3, X, Y, Z, nt, _lib.ptr(ijk), None, 0.0, mask,
                                                              _lib.ptr(feat), ld, None, 0, None, None, None, st)
    assert ok(1) == 0
    assert ok(0) == -1 and b"num_targets" in lib.rml_last_error()
    assert ok(5) == -1                                                   # more targets than X
    assert ok(1, mask=0) == -1 and b"mask" in lib.rml_last_error()
    assert ok(1, ld=D - 1) == -1 and b"ld_feat" in lib.rml_last_error()
    assert ok(1, vol=None) == -1
    assert lib.rml_derive_slice(ctx, None, 0, 0, X, Y, Z, 1, None, None, 0.0, 7, None, 0, None, 0, None, None, None, st) == 0      # B == 0
    assert lib.rml_derive_slice(ctx, _lib.ptr(v), 7, 3, X, Y, Z, 1, None, None, 0.0, 7, _lib.ptr(feat), D, None, 0, None, None, None, st) == -1
    # support predicate: whole quads, Z <= 256, odd part of Z/4 <= 15, aligned base
    assert lib.rml_derive_slice_supported(ctx, None, 0, 22, 31, 176, 1) == 1 and lib.rml_derive_slice_supported(None, None, 1, 64, 64, 128, 3) == 1
    assert lib.rml_derive_slice_supported(ctx, None, 0, 4, 4, 18, 1) == 0 and lib.rml_derive_slice_supported(ctx, None, 0, 4, 4, 132, 1) == 0
    assert lib.rml_derive_slice_supported(ctx, None, 0, 4, 4, 260, 1) == 0 and lib.rml_derive_slice_supported(ctx, None, 9, 4, 4, 16, 1) == 0
    assert lib.rml_derive_slice_supported(ctx, _lib.c_void_p(v.data_ptr() + 4), 0, X, Y, Z, 1) == 0
    # rml_derive_project_svm refuses a shape without the fused kernel with a message that names the way out
    import numpy as np
    sv = np.zeros((4, 4 * 18 + 4 * 18 + 16)); dc = np.zeros((2, 4)); ic = np.zeros(3); ns = np.array([2, 1, 1], dtype=np.int32)
    h = C.c_void_p()
    assert lib.rml_svm_load(ctx, sv.ctypes.data, 4, sv.shape[1], dc.ctypes.data, ic.ctypes.data, ns.ctypes.data, 3, 0, 0.1, 255.0, None, None, C.byref(h)) == 0
    v18 = torch.zeros((2, 4, 4, 18), device="cuda")
    lab = torch.empty((2,), dtype=torch.int32, device="cuda")
    assert lib.rml_derive_project_svm(ctx, h, _lib.ptr(v18), 0, 2, 4, 4, 18, 255.0, 7, None, None, None, None, _lib.ptr(lab), None, st) == -2
    assert b"rml_derive_targets" in lib.rml_last_error()
    lib.rml_svm_free(ctx, h)
    # options: every id of radarml.h round-trips through set / get, values are normalised, unknown ids and bad values are refused
    assert lib.rml_ctx_set_option(ctx, 99, 1) == -1 and b"option" in lib.rml_last_error()
    assert lib.rml_ctx_set_option(None, _lib.OPT_PROJECT_SHARE_CU, 1) == -1
    got = C.c_int()
    assert lib.rml_ctx_get_option(ctx, 99, C.byref(got)) == -1 and lib.rml_ctx_get_option(ctx, _lib.OPT_CHUNK, None) == -1
    defaults = {"project_share_cu": 0, "waveframe": 1, "linplane": 1, "stage_codes": 1, "slice_wave": 1, "derive_fused": 1, "code_rmw": -1,
                "gemm_big": -1, "chunk": 0, "c1_pk": 1}
    assert set(defaults) == set(_lib.OPTIONS)
    for name, dflt in defaults.items():
        assert _lib.get_option(name) == dflt, name
    with _lib.options(waveframe=2, gemm_big=5, code_rmw=-7, chunk=4096, linplane=0):
        assert (_lib.get_option("waveframe"), _lib.get_option("gemm_big"), _lib.get_option("code_rmw"), _lib.get_option("chunk"), _lib.get_option("linplane")) == (2, 1, -1, 4096, 0)
    for name, dflt in defaults.items():
        assert _lib.get_option(name) == dflt, name
    assert lib.rml_ctx_set_option(ctx, _lib.OPT_WAVEFRAME, 4) == -1 and lib.rml_ctx_set_option(ctx, _lib.OPT_CHUNK, 64) == -1
    assert lib.rml_ctx_set_option(ctx, _lib.OPT_DERIVE_FUSED, 0) == 0
    try:
        assert lib.rml_derive_slice_supported(ctx, None, 0, 22, 31, 176, 1) == 0 and lib.rml_derive_slice_supported(None, None, 0, 22, 31, 176, 1) == 1
    finally:
        assert lib.rml_ctx_set_option(ctx, _lib.OPT_DERIVE_FUSED, 1) == 0
    V, _ = rml.synth_volumes(1500, 22, 31, 176, seed=2)
    a = rml.process_volumes(V, mode="max", scale=True)
    assert lib.rml_ctx_set_option(ctx, _lib.OPT_PROJECT_SHARE_CU, 1) == 0
    try:
        b = rml.process_volumes(V, mode="max", scale=True)
    finally:
        assert lib.rml_ctx_set_option(ctx, _lib.OPT_PROJECT_SHARE_CU, 0) == 0
    assert torch.equal(a, b)


def test_cnn_chain_entry_points_reject_bad_arguments(rml):
    """rml_dnn_preprocess_supported / _rows / _volumes, rml_dnn_trunk_kblock, rml_dnn_dense_workspace_bytes / rml_dnn_dense_tail
    through the C ABI: status codes and messages for bad arguments, the support predicate, the B == 0 cases."""
    import torch
    from radar_ml_amd import _lib
    lib = _lib.load()
    ctx = _lib.context()
    st = _lib.stream_ptr()
    # support predicate: Z % 16 == 0, out_w % 4 == 0 and <= 256, no vertical shrink, rows of at most 20 480 elements
    assert lib.rml_dnn_preprocess_supported(22, 31, 176, 80, 80) == 1 and lib.rml_dnn_preprocess_supported(64, 64, 128, 80, 80) == 1
    assert lib.rml_dnn_preprocess_supported(22, 31, 180, 80, 80) == 0           # Z % 16
    assert lib.rml_dnn_preprocess_supported(22, 31, 176, 80, 82) == 0           # out_w % 4
    assert lib.rml_dnn_preprocess_supported(22, 31, 176, 16, 80) == 0           # the height would shrink: windows of more than 4 taps
    assert lib.rml_dnn_preprocess_supported(64, 64, 256, 80, 80) == 0           # row longer than 20 480
    assert lib.rml_dnn_preprocess_supported(0, 31, 176, 80, 80) == 0
    X, Y, Z = 5, 7, 16
    D = X * Z + Y * Z + X * Y
    B = 3
    ldq = 256
    feat = torch.zeros((B, D), device="cuda")
    codes = torch.zeros((B, ldq), dtype=torch.uint8, device="cuda")
    flags = torch.ones((B + 1,), dtype=torch.int32, device="cuda")
    outs = [torch.empty((B, 12, 16), dtype=torch.bfloat16, device="cuda") for _ in range(3)]
    rows = lambda f=feat, ld=D, c=None, lq=0, fl=None, o0=outs[0], ow=16: lib.rml_dnn_preprocess_rows(
        ctx, _lib.ptr(f), ld, _lib.ptr(c), lq, _lib.ptr(fl), B, X, Y, Z, 12, ow, _lib.ptr(o0), _lib.ptr(outs[1]), _lib.ptr(outs[2]), st)
    assert rows() == 0
    assert rows(c=codes, lq=ldq, fl=flags) == 0
    assert rows(f=None) == -1 and b"NULL" in lib.rml_last_error()
    assert rows(ld=D - 1) == -1 and b"ld" in lib.rml_last_error()
    assert rows(c=codes, lq=ldq) == -1 and b"flags" in lib.rml_last_error()     # both kinds of rows, nothing to choose by
    assert rows(f=None, c=codes, lq=D) == -1 and b"ldq" in lib.rml_last_error()    # ldq % 16
    assert rows(o0=None) == -1
    assert rows(ow=18) == -2 and b"rml_resize_bicubic" in lib.rml_last_error()  # unsupported shape names the way out
    assert lib.rml_dnn_preprocess_rows(ctx, None, 0, None, 0, None, 0, X, Y, Z, 12, 16, None, None, None, st) == 0      # B == 0
    v = torch.zeros((B, X, Y, Z), device="cuda")
    scratch = torch.empty((B, D), device="cuda")
    vols = lambda vol=v, vdt=0, c=codes, sc=scratch, lq=ldq: lib.rml_dnn_preprocess_volumes(
        ctx, _lib.ptr(vol), vdt, B, X, Y, Z, 0, None, _lib.ptr(c), lq, _lib.ptr(flags), _lib.ptr(sc), D, 12, 16,
        _lib.ptr(outs[0]), _lib.ptr(outs[1]), _lib.ptr(outs[2]), st)
    assert vols() == 0
    assert vols(vol=None) == -1 and vols(c=None) == -1
    assert vols(vdt=5) == -1 and b"dtype" in lib.rml_last_error()
    assert vols(sc=None) == -1 and b"scratch" in lib.rml_last_error()           # float32 volumes need the float-row scratch ...
    assert vols(vol=v.to(torch.uint8), vdt=1, sc=None) == 0                     # ... uint8 volumes do not
    assert vols(lq=D + 1) == -1
    torch.cuda.synchronize()
    # the dense tail
    K, N = 38400, 5
    fv = torch.zeros((N, K), dtype=torch.bfloat16, device="cuda")
    w1 = torch.zeros((64, K), dtype=torch.bfloat16, device="cuda")
    b64 = torch.zeros(64, device="cuda"); w2t = torch.zeros((64, 64), device="cuda"); w3 = torch.zeros((3, 64), device="cuda"); b3 = torch.zeros(3, device="cuda")
    nbytes = lib.rml_dnn_dense_workspace_bytes(ctx, N, K)
    assert nbytes >= N * 64 * 4 and lib.rml_dnn_dense_workspace_bytes(ctx, 0, K) == 0
    ws = torch.empty((nbytes // 4,), device="cuda")
    pr = torch.empty((N, 3), device="cuda")
    tail = lambda k=K, f=fv, nb=nbytes, nc=3, kb=0: lib.rml_dnn_dense_tail(ctx, _lib.ptr(f), K, kb, N, k, _lib.ptr(w1), _lib.ptr(b64), _lib.ptr(w2t),
                                                                             _lib.ptr(b64), _lib.ptr(w3), _lib.ptr(b3), nc, _lib.ptr(ws), nb, _lib.ptr(pr), st)
    assert tail() == 0
    torch.cuda.synchronize()
    assert torch.allclose(pr, torch.full_like(pr, 1.0 / 3.0))                   # zero weights: uniform probabilities
    assert tail(k=K - 32) == -2 and b"multiple of 64" in lib.rml_last_error()
    assert tail(f=None) == -1 and tail(nb=nbytes - 4) == -1 and b"workspace" in lib.rml_last_error()
    assert tail(nc=17) == -2
    # the K-block trunk needs an even number of output pixels
    x = torch.zeros((2, 12, 8), dtype=torch.bfloat16, device="cuda")           # 3 x 2 = 6 output pixels: fine; 12 x 8 planes
    w1c = torch.zeros((3, 64, 9), device="cuda"); b1c = torch.zeros((3, 64), device="cuda")
    w2c = torch.zeros((3, 32, 576), dtype=torch.bfloat16, device="cuda"); b2c = torch.zeros((3, 32), device="cuda")
    fk = torch.empty((6 * 96 // 64, 2, 64), dtype=torch.bfloat16, device="cuda")
    kb = lambda H, W: lib.rml_dnn_trunk_kblock(ctx, _lib.ptr(x), _lib.ptr(x), _lib.ptr(x), 1, 2, H, W, _lib.ptr(w1c), _lib.ptr(b1c), _lib.ptr(w2c),
                                               _lib.ptr(b2c), _lib.ptr(fk), st)
    assert kb(12, 8) == 0
    torch.cuda.synchronize()
    # bf16 planes (W % 8 == 0) always have an even number of output pixels; float32 planes need not: 12 x 12 -> 3 x 3
    xf = torch.zeros((2, 12, 12), device="cuda")
    assert lib.rml_dnn_trunk_kblock(ctx, _lib.ptr(xf), _lib.ptr(xf), _lib.ptr(xf), 0, 2, 12, 12, _lib.ptr(w1c), _lib.ptr(b1c), _lib.ptr(w2c),
                                    _lib.ptr(b2c), _lib.ptr(fk), st) == -2 and b"even" in lib.rml_last_error()


def test_probe_stream_and_rmw_default(rml):
    """rml_probe_stream: the streaming-read denominator of bench.py -- argument checks and a plausible MI355X figure (between a
    third of the 8 TB/s specification and the specification); rml_code_rmw_default follows the rule, whatever the environment says."""
    import os
    import torch
    from radar_ml_amd import _lib
    lib = _lib.load()
    ctx = _lib.context()
    st = _lib.stream_ptr()
    gbs = C.c_double()
    buf = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")
    assert lib.rml_probe_stream(ctx, None, buf.numel(), 3, C.byref(gbs), st) == -1
    assert lib.rml_probe_stream(ctx, _lib.ptr(buf), 1000, 3, C.byref(gbs), st) == -1 and b"1 MiB" in lib.rml_last_error()
    assert lib.rml_probe_stream(ctx, C.c_void_p(buf.data_ptr() + 4), buf.numel() - 16, 3, C.byref(gbs), st) == -1
    assert lib.rml_probe_stream(ctx, _lib.ptr(buf), buf.numel(), 10, C.byref(gbs), st) == 0
    print("rml_probe_stream: %.0f GB/s" % gbs.value)
    # a sanity bound that holds on any HIP device (an MI355X reads 6.3-6.9 TB/s here; a busy box or another part reads less)
    assert 100.0 < gbs.value < 16000.0
    # 64x64x128: uint8 volumes on, float32 max off, derive off (< 2 % on fresh frames); Walabot grid uint8 off (csrc/rml_internal.h).
    # The rule itself: nothing in the process environment moves it (RML_OPT_CODE_RMW overrides it per context, in the pipelines)
    os.environ["RML_CODE_RMW"] = "1"
    try:
        assert lib.rml_code_rmw_default(20480, 64 * 64 * 128, 0, 1) == 1
        assert lib.rml_code_rmw_default(20480, 4 * 64 * 64 * 128, 0, 0) == 0
        assert lib.rml_code_rmw_default(20480, 4 * 64 * 64 * 128, 1, 0) == 0
        assert lib.rml_code_rmw_default(10010, 22 * 31 * 176, 0, 1) == 0
    finally:
        os.environ.pop("RML_CODE_RMW", None)


def test_four_threads_two_streams_one_context(rml):
    """one rml_ctx shared by host threads -- the workspace users serialise on the context's mutex and on its
    last-use event.  Four threads over two streams hammer the three workspace users at once (the fused projection -> SVM pipeline
    with several chunks, rml_svm_decision on rows, derive + projection); every result must be the single-threaded one, bit for
    bit, every time."""
    import threading
    import torch
    import oracle_np as O
    from conftest import load_golden, svm_model_arrays
    g = load_golden("svm_walabot.npz")
    sv = (g["sv_u8"].astype(np.float32) / np.float32(255.0)).astype(np.float64)
    svc = rml.GpuSVC(sv, g["dual_coef"], g["intercept"], g["n_support"], float(g["gamma"]), g["classes"],
                     calib_a=g["calib_a"], calib_b=g["calib_b"])
    vol = torch.from_numpy(g["test_vol_u8"].astype(np.float32)).cuda()                  # (128, 22, 31, 176)
    big = vol.repeat(80, 1, 1, 1)                                                        # 10 240 frames: two pipeline chunks
    rows = rml.process_volumes(vol, mode="max", scale=True)
    want_pipe = {k: v.clone() for k, v in svc.decide_volumes(big, mode="max", scale=True, want_proba=True).items()}
    want_rows = svc.decision_function(rows.cpu().numpy())
    want_ijk = rml.derive_targets(vol, 2).cpu().numpy()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    errs = []

    def work(i):
        try:
            with torch.cuda.stream(streams[i % 2]):
                for it in range(12):
                    if i in (0, 1):
                        out = svc.decide_volumes(big, mode="max", scale=True, want_proba=True)
                        streams[i % 2].synchronize()
                        for k in ("dec_ovo", "proba", "label_vote", "label_calib"):
                            assert torch.equal(out[k], want_pipe[k]), (i, it, k)
                    elif i == 2:
                        assert np.array_equal(svc.decision_function(rows.cpu().numpy()), want_rows), (i, it)
                    else:
                        assert np.array_equal(rml.derive_targets(vol, 2).cpu().numpy(), want_ijk), (i, it)
                        p = rml.process_volumes(vol, mode="max", scale=True)
                        assert torch.equal(p, rows), (i, it)
        except Exception as e:          # pragma: no cover
            errs.append(repr(e)[:500])

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not any(t.is_alive() for t in ts), "a thread is stuck"
    assert not errs, errs


def test_c_driver_four_threads_on_one_context(rml):
    """The sanitizer driver's device phase (tools/sanitize/driver.cpp), as a plain C++ program against the real library: no Python,
    no torch allocator -- hipMalloc'ed buffers, four host threads on two streams through rml_project_svm on one rml_ctx, results
    compared bit for bit.  (Its host phase runs under ASAN/UBSAN in the CPU suite.)"""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    from radar_ml_amd import _lib
    sys.path.insert(0, os.path.join(ROOT, "tools", "sanitize"))
    import importlib
    sb = importlib.import_module("build")
    sys.path.pop(0)
    drv = sb.build_driver_for(_lib.LIB_PATH)
    r = subprocess.run([drv, "--need-device"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, cwd="/tmp")
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-4000:]
    assert "phase A: ok" in out and "phase B: 9000 frames x 4 threads" in out and "ALL OK" in out, out[-4000:]


def test_entry_points_capture_into_a_hip_graph_after_a_warm_up(rml):
    """include/radarml.h: "launches are asynchronous on the caller's stream" -- so a warmed-up call (workspace grown, tables cached)
    records into a HIP graph: rml_project_svm (its second stream and events join the capture) and the CNN chain
    (rml_dnn_preprocess_volumes, rml_dnn_trunk_kblock, rml_dnn_dense_tail).  Replays on new frames written into the captured
    buffers give the eager results bit for bit; a call that would have to GROW the workspace inside a capture is refused with a
    message that names rml_ctx_reserve_workspace, and the context works normally afterwards."""
    import importlib
    import torch
    from conftest import load_golden, svm_model_arrays
    from radar_ml_amd import _lib
    lib = _lib.load()
    ctx = _lib.context()
    dnn = importlib.import_module("radar_ml_amd.dnn")
    g = load_golden("svm_walabot.npz")
    m = svm_model_arrays(g)
    svc = rml.GpuSVC(m["sv"], m["dual_coef"], m["intercept"], m["n_support"], m["gamma"], m["classes"], calib_a=m["calib_a"], calib_b=m["calib_b"])
    torch.manual_seed(3)
    model = dnn.define_classifier(device="cuda").eval()
    B = 3000
    Va, _ = rml.synth_volumes(B, 22, 31, 176, seed=41)
    Vb, _ = rml.synth_volumes(B, 22, 31, 176, seed=43)
    eager = {}
    for name, V in (("a", Va), ("b", Vb)):
        o = svc.decide_volumes(V, mode="max", scale=True, want_proba=True)
        eager[name] = (o["dec_ovo"].clone(), o["label_calib"].clone(), model.predict_volumes(V, label_guard=None).clone())
    assert not torch.equal(eager["a"][0], eager["b"][0])
    buf = Va.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # the warm-up on the capturing stream (torch's recipe)
        svc.decide_volumes(buf, mode="max", scale=True, want_proba=True)
        model.predict_volumes(buf, label_guard=None)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        o = svc.decide_volumes(buf, mode="max", scale=True, want_proba=True)
        p = model.predict_volumes(buf, label_guard=None)
    for name, V in (("b", Vb), ("a", Va), ("b", Vb)):
        buf.copy_(V)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(o["dec_ovo"], eager[name][0]) and torch.equal(o["label_calib"], eager[name][1]), name
        assert torch.equal(p, eager[name][2]), name
    # the single-observation and the split-K paths (one stream, no events; a memset node in the second) capture and replay as well
    small = {}
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for nb in (1, 64):
            svc.decide_volumes(buf[:nb], mode="max", scale=True, want_proba=True)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph_s = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph_s, stream=side):
        for nb in (1, 64):
            small[nb] = svc.decide_volumes(buf[:nb], mode="max", scale=True, want_proba=True)
    for name, V in (("a", Va), ("b", Vb)):
        buf.copy_(V)
        graph_s.replay()
        torch.cuda.synchronize()
        for nb in (1, 64):
            assert torch.equal(small[nb]["dec_ovo"], eager[name][0][:nb]) and torch.equal(small[nb]["label_calib"], eager[name][1][:nb]), (name, nb)
    # growth inside a capture: refused, nothing broken
    have = int(lib.rml_ctx_workspace_bytes(ctx))
    assert have > 0
    big, _ = rml.synth_volumes(9000, 64, 64, 128, seed=5)          # chunks of 64x64x128 code rows need more than the Walabot ones did
    rng = np.random.default_rng(8)
    D64 = 64 * 128 * 2 + 64 * 64
    sv64 = (rng.integers(0, 256, (256, D64)).astype(np.float32) / np.float32(255.0)).astype(np.float64)
    svc64 = rml.GpuSVC(sv64, rng.normal(size=(2, 256)), np.zeros(3), np.array([86, 85, 85], dtype=np.int32), 0.01, np.arange(3), device="cuda")
    # (the context is the process's: when tests that ran BEFORE this one -- another file order than the suite's -- have already grown
    # its workspace past what these chunks need, there is no growth to refuse)
    if have < (3 << 29):
        graph2 = torch.cuda.CUDAGraph()
        with pytest.raises(Exception) as ei:
            with torch.cuda.graph(graph2, stream=side):
                svc64.decide_volumes(big, mode="max", scale=True)
        assert "rml_ctx_reserve_workspace" in str(ei.value)
        torch.cuda.synchronize()
        assert int(lib.rml_ctx_workspace_bytes(ctx)) == have
    svc64.decide_volumes(big, mode="max", scale=True)               # un-captured: grows (the outgrown block is parked, then freed)
    torch.cuda.synchronize()
    have = int(lib.rml_ctx_workspace_bytes(ctx))
    # explicit reservation: grows (synchronising), never shrinks
    assert lib.rml_ctx_reserve_workspace(ctx, have + (64 << 20)) == 0
    assert int(lib.rml_ctx_workspace_bytes(ctx)) >= have + (64 << 20)
    assert lib.rml_ctx_reserve_workspace(ctx, 1024) == 0 and int(lib.rml_ctx_workspace_bytes(ctx)) >= have + (64 << 20)
    assert lib.rml_ctx_reserve_workspace(None, 1024) == -1
    o2 = svc.decide_volumes(Va, mode="max", scale=True, want_proba=True)
    assert torch.equal(o2["dec_ovo"], eager["a"][0])
