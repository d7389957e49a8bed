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
