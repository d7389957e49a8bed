"""ctypes wrapper of the C oracle (oracle/oracle.c).  TEST INFRASTRUCTURE ONLY -- see oracle_np.py."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            import importlib.util
            spec = importlib.util.spec_from_file_location("oracle_build", os.path.join(HERE, "build.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            path = mod.build()
        _lib = C.CDLL(path)
        _lib.oracle_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def project_max(V, threads=1):
    V = np.ascontiguousarray(V, dtype=np.float32)
    B, X, Y, Z = V.shape
    xz = np.empty((B, X, Z), np.float32); yz = np.empty((B, Y, Z), np.float32); xy = np.empty((B, X, Y), np.float32)
    lib().oracle_project_max(_p(V), C.c_int64(B), X, Y, Z, _p(xz), _p(yz), _p(xy), int(threads))
    return xz, yz, xy


def project_slice(V, ijk):
    V = np.ascontiguousarray(V, dtype=np.float32)
    ijk = np.ascontiguousarray(ijk, dtype=np.int32)
    B, X, Y, Z = V.shape
    xz = np.empty((B, X, Z), np.float32); yz = np.empty((B, Y, Z), np.float32); xy = np.empty((B, X, Y), np.float32)
    lib().oracle_project_slice(_p(V), C.c_int64(B), X, Y, Z, _p(ijk), _p(xz), _p(yz), _p(xy))
    return xz, yz, xy


def features(xz, yz, xy, mask=(True, True, True), scale=False):
    B, X, Z = xz.shape
    Y = yz.shape[1]
    bits = sum(1 << i for i in range(3) if mask[i])
    D = (X * Z if mask[0] else 0) + (Y * Z if mask[1] else 0) + (X * Y if mask[2] else 0)
    f = np.empty((B, D), np.float32)
    lib().oracle_features(_p(np.ascontiguousarray(xz)), _p(np.ascontiguousarray(yz)), _p(np.ascontiguousarray(xy)),
                          C.c_int64(B), X, Y, Z, bits, int(bool(scale)), _p(f))
    return f


def svm(X, sv, dual_coef, intercept, n_support, gamma, kernel="rbf", calib_a=None, calib_b=None, threads=1):
    """Returns dict(dec_ovo, dec_ovr, label_vote[, proba, label_calib]) -- class INDICES for labels."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    sv = np.ascontiguousarray(sv, dtype=np.float64)
    dc = np.ascontiguousarray(dual_coef, dtype=np.float64)
    ic = np.ascontiguousarray(intercept, dtype=np.float64)
    ns = np.ascontiguousarray(n_support, dtype=np.int32)
    N, D = X.shape
    M = sv.shape[0]
    Cn = ns.shape[0]
    P = Cn * (Cn - 1) // 2
    out = {"dec_ovo": np.empty((N, P)), "dec_ovr": np.empty((N,) if Cn == 2 else (N, Cn)),
           "label_vote": np.empty((N,), np.int32)}
    ca = cb = None
    if calib_a is not None:
        ca = np.ascontiguousarray(calib_a, dtype=np.float64); cb = np.ascontiguousarray(calib_b, dtype=np.float64)
        out["proba"] = np.empty((N, Cn)); out["label_calib"] = np.empty((N,), np.int32)
    lib().oracle_svm(_p(X), C.c_int64(N), C.c_int64(D), _p(sv), C.c_int64(M), _p(dc), _p(ic), _p(ns), Cn,
                     0 if kernel == "rbf" else 1, C.c_double(gamma), _p(ca), _p(cb),
                     _p(out["dec_ovo"]), _p(out["dec_ovr"]), _p(out.get("proba")), _p(out["label_vote"]),
                     _p(out.get("label_calib")), int(threads))
    return out


def max_threads():
    return int(lib().oracle_max_threads())
