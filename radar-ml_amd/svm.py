"""sklearn-protocol classifiers backed by the HIP SVM kernels.

The reference pickles a fitted ``CalibratedClassifierCV`` wrapping an ``SVC`` (or an
``SGDClassifier``) and calls ``.predict`` / ``.predict_proba`` on it (train.py:217,722-731;
predict.py:60).  ``from_sklearn(obj)`` ingests such a fitted object and returns an object with
the same protocol (``predict``, ``predict_proba``, ``decision_function``, ``classes_``) whose
arithmetic runs on the GPU.  Fitting stays on the CPU with scikit-learn.
"""
import ctypes as C

import numpy as np

from . import _lib
from .common import ProjMask, RADAR_MAX, _as_device_f32, _as_device_volumes, _mask_bits


def _torch():
    import torch
    return torch


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _detect_code_scale(sv):
    """255.0 when every SV is bit-identical to float32(c/255) (train.py:667 scaling), 1.0 when it
    is an integer code itself, else 0.0 (general float path only)."""
    sv = np.asarray(sv, dtype=np.float64)
    for scale in (255.0, 1.0):
        c = np.rint(sv * scale)
        if c.min() < 0 or c.max() > 255:
            continue
        back = (c.astype(np.float32) / np.float32(scale)).astype(np.float64) if scale > 1 else c
        if np.array_equal(back, sv):
            return scale
    return 0.0


class _Base:
    classes_ = None

    def _rows(self, X, check_finite=True):
        """Accept numpy or torch, any real dtype -> (N,D) float32 CUDA tensor.  ``check_finite`` (the sklearn-protocol methods):
        scikit-learn's own input validation -- ``clf.predict`` of predict.py:60 goes through ``validate_data`` and raises
        ValueError on a NaN or an infinity; so do the GPU twins (one reduction pass over the rows on the device; these methods
        hand NumPy arrays back and synchronise anyway).  The tensor-in / tensor-out entries (``_decide``, ``decide_volumes``) stay
        asynchronous and do not check."""
        torch = _torch()
        if not isinstance(X, torch.Tensor):
            X = np.asarray(X)
        if X.ndim != 2:
            raise ValueError("Expected 2D array, got %dD array instead" % X.ndim)
        if X.shape[1] != self.n_features_in_:
            raise ValueError("X has %d features, but %s is expecting %d features as input."
                             % (X.shape[1], type(self).__name__, self.n_features_in_))
        # rows go to the model's device (its buffers and its context live there), whatever the current device is
        host = not isinstance(X, torch.Tensor)
        if check_finite and host and 0 < X.size <= (1 << 22):
            # a host array of a few rows (predict.py:60: ONE observation per call): validated where scikit-learn validates it, on the
            # host, before the upload -- the device pass below costs five small launches and a synchronisation (~40 us of a ~140 us call)
            if not bool(np.isfinite(X).all()):
                raise ValueError("Input contains NaN, infinity or a value too large for dtype('float64').")      # sklearn's message
            check_finite = False
        Xd = _as_device_f32(X, getattr(self, "_dev", None))
        if check_finite and Xd.numel() and not bool(torch.isfinite(Xd).all()):
            raise ValueError("Input contains NaN, infinity or a value too large for dtype('float64').")      # sklearn's message
        return Xd


MAX_CLASSES = 6         # kMaxC of csrc/svm.hip (15 one-vs-one pairs)


def _check_classes(n):
    if not 2 <= n <= MAX_CLASSES:
        raise NotImplementedError("%d classes: the HIP SVM / linear kernels are built for 2..%d classes "
                                  "(csrc/svm.hip kMaxC)" % (n, MAX_CLASSES))


class GpuSVC(_Base):
    """GPU twin of a fitted ``sklearn.svm.SVC`` (C-SVC, one-vs-one, RBF or linear kernel)."""

    def __init__(self, support_vectors, dual_coef, intercept, n_support, gamma, classes, kernel="rbf",
                 calib_a=None, calib_b=None, decision_function_shape="ovr", device=None, path="auto",
                 probA=None, probB=None):
        torch = _torch()
        lib = _lib.load()
        sv = _f64(support_vectors)
        self.classes_ = np.asarray(classes)
        self.n_features_in_ = sv.shape[1]
        self.decision_function_shape = decision_function_shape
        self.kernel = kernel
        self.gamma = float(gamma)
        self.path = path
        C_ = len(self.classes_)
        _check_classes(C_)
        dc, ic, ns = _f64(dual_coef), _f64(intercept), np.ascontiguousarray(n_support, dtype=np.int32)
        if dc.shape != (C_ - 1, sv.shape[0]):
            raise ValueError("dual_coef must be (n_classes-1, n_SV) in libsvm order (SVC._dual_coef_)")
        if not torch.cuda.is_available():
            raise _lib.RadarMLError("no HIP device is visible: the radar-ml HIP path needs an MI355X (no CPU fallback)")
        self._dev = _lib.device_of(device)
        self._ctx = _lib.context(self._dev)
        self.code_scale = _detect_code_scale(sv)
        ca = cb = None
        if calib_a is not None:
            ca, cb = _f64(calib_a), _f64(calib_b)
        h = C.c_void_p()
        with torch.cuda.device(self._dev):
            _lib.check(lib.rml_svm_load(
                self._ctx, sv.ctypes.data, sv.shape[0], sv.shape[1], dc.ctypes.data, ic.ctypes.data, ns.ctypes.data, C_,
                _lib.KERNEL_RBF if kernel == "rbf" else _lib.KERNEL_LINEAR, float(gamma),
                self.code_scale if self.code_scale > 0 else 1.0,
                ca.ctypes.data if ca is not None else None, cb.ctypes.data if cb is not None else None, C.byref(h)),
                "rml_svm_load")
        self._h = h
        self.has_calibration = ca is not None
        # libsvm's own Platt coefficients (SVC(probability=True)): used by predict_proba of the bare SVC
        self._probA = None if probA is None or len(np.ravel(probA)) == 0 else _f64(np.ravel(probA))
        self._probB = None if probB is None or len(np.ravel(probB)) == 0 else _f64(np.ravel(probB))
        if self._probA is not None:
            P = C_ * (C_ - 1) // 2
            if len(self._probA) != P or self._probB is None or len(self._probB) != P:
                raise ValueError("probA / probB must hold one value per class pair (%d)" % P)
            with torch.cuda.device(self._dev):      # uploaded once: predict_proba is then an asynchronous launch
                _lib.check(lib.rml_svm_set_platt(self._ctx, h, self._probA.ctypes.data, self._probB.ctypes.data), "rml_svm_set_platt")
        self.exact = bool(lib.rml_svm_is_exact(h)) and self.code_scale > 0
        self.n_sv = sv.shape[0]

    # -- construction from scikit-learn -----------------------------------------------------
    @classmethod
    def from_sklearn(cls, clf, calib=None, **kw):
        """clf: fitted sklearn.svm.SVC.  calib: optional (a, b) arrays of the sigmoid calibrators."""
        kernel = clf.kernel
        if kernel not in ("rbf", "linear"):
            raise NotImplementedError("SVC kernel %r (the reference's grid uses linear and rbf: train.py:474-476)" % kernel)
        a = b = None
        if calib is not None:
            a, b = calib
        # private libsvm-order arrays; the fall-backs are the names older pickles carry (the reference pins scikit-learn
        # 0.24, requirements.txt:57: probA_ / probB_ are plain attributes there, n_support_ a property of _n_support)
        def pick(*names):
            for nm in names:
                v = clf.__dict__.get(nm, None) if nm.startswith("_") else getattr(clf, nm, None)
                if v is not None:
                    return v
            return None
        dual = pick("_dual_coef_", "dual_coef_")
        icpt = pick("_intercept_", "intercept_")
        if "_dual_coef_" not in clf.__dict__ and len(clf.classes_) == 2:
            dual, icpt = -np.asarray(dual), -np.asarray(icpt)       # sk:svm/_base.py:266-270: the public pair is negated
        return cls(clf.support_vectors_, dual, icpt, pick("_n_support", "n_support_"), pick("_gamma", "gamma"), clf.classes_,
                   kernel=kernel, calib_a=a, calib_b=b, probA=pick("_probA", "probA_"), probB=pick("_probB", "probB_"),
                   decision_function_shape=getattr(clf, "decision_function_shape", "ovr"), **kw)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().rml_svm_free(self._ctx, self._h)
                self._h = None
        except Exception:
            pass

    # -- raw device call --------------------------------------------------------------------
    def _decide(self, Xd, want_proba=False, path=None):
        torch = _torch()
        lib = _lib.load()
        N = Xd.shape[0]
        C_ = len(self.classes_)
        P = C_ * (C_ - 1) // 2
        dev = Xd.device
        ovo = torch.empty((N, P), dtype=torch.float64, device=dev)
        ovr = torch.empty((N,) if C_ == 2 else (N, C_), dtype=torch.float64, device=dev)
        vote = torch.empty((N,), dtype=torch.int32, device=dev)
        proba = lab = None
        if want_proba:
            proba = torch.empty((N, C_), dtype=torch.float64, device=dev)
            lab = torch.empty((N,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.rml_svm_decision(
                self._ctx, self._h, _lib.PATHS[path or self.path], _lib.ptr(Xd), Xd.stride(0), None, 0, None, None, None, N,
                _lib.ptr(ovo), _lib.ptr(ovr), _lib.ptr(proba), _lib.ptr(vote), _lib.ptr(lab), _lib.stream_ptr(dev)),
                "rml_svm_decision")
        return ovo, ovr, vote, proba, lab

    def decide_codes(self, codes, row_isum, row_isq, row_flags=None, want_proba=False):
        """Decision from biased uint8 code rows as ``process_volumes(..., codes=True)`` returns them (exact path only:
        no row preparation, the GEMM + finish kernels alone).  Returns (dec_ovo, dec_ovr, label_vote, proba, label_calib)."""
        torch = _torch()
        lib = _lib.load()
        N = codes.shape[0]
        C_ = len(self.classes_)
        P = C_ * (C_ - 1) // 2
        dev = codes.device
        ovo = torch.empty((N, P), dtype=torch.float64, device=dev)
        ovr = torch.empty((N,) if C_ == 2 else (N, C_), dtype=torch.float64, device=dev)
        vote = torch.empty((N,), dtype=torch.int32, device=dev)
        proba = lab = None
        if want_proba:
            proba = torch.empty((N, C_), dtype=torch.float64, device=dev)
            lab = torch.empty((N,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.rml_svm_decision(
                self._ctx, self._h, _lib.PATH_I8, None, 0, _lib.ptr(codes), codes.stride(0), _lib.ptr(row_isum), _lib.ptr(row_isq),
                _lib.ptr(row_flags), N, _lib.ptr(ovo), _lib.ptr(ovr), _lib.ptr(proba), _lib.ptr(vote), _lib.ptr(lab),
                _lib.stream_ptr(dev)), "rml_svm_decision")
        return ovo, ovr, vote, proba, lab

    def decide_volumes(self, volumes, mode="max", ijk=None, proj_mask=ProjMask(True, True, True), scale=True,
                       want_proba=None, validate_ijk=True):
        """Fused batched path: (B,X,Y,Z) volumes -> projection -> SVM, features never leave the GPU.
        Returns a dict of CUDA tensors (dec_ovo, dec_ovr, label_vote[, proba, label_calib]); mode='slice' without ``ijk``
        slices through the strongest derived target of every frame (common.py:49-80) and also returns it as ``ijk`` (B,3)
        where the one-pass kernel ran.  ``validate_ijk=False``: the caller vouches that a DEVICE ``ijk`` is within
        [-size, size) -- the range check of a device tensor costs a synchronising read-back per call."""
        torch = _torch()
        lib = _lib.load()
        from .common import derive_targets, _slice_indices, process_volumes
        v, vdt = _as_device_volumes(volumes, self._dev)
        if v.ndim == 3:
            v = v.unsqueeze(0)
        B, X, Y, Z = v.shape
        dev = v.device
        C_ = len(self.classes_)
        P = C_ * (C_ - 1) // 2
        if want_proba is None:
            want_proba = self.has_calibration
        ijk_t = None
        fused_derive = False
        if mode == "slice":
            derived = ijk is None
            # no SDK target: DerivedTarget.get_derived_targets (common.py:49-80) on the GPU -- in the same pass as the slices and
            # the SVM where the shape has the fused kernel (rml_derive_project_svm), as a launch of its own otherwise
            fused_derive = derived and bool(lib.rml_derive_slice_supported(_lib.context(v.device), _lib.ptr(v), vdt, X, Y, Z, 1))
            if derived and not fused_derive:
                ijk = derive_targets(v, 1)[:, 0, :]
            if fused_derive:
                T = 1
            else:
                ijk_t, T = _slice_indices(ijk, B, X, Y, Z, dev, validate=validate_ijk and not derived)
            if T > 1:
                # several targets per frame (predict.py:93-119 classifies every target of one image): slice rows first,
                # then the SVM on the B*T rows; outputs are (B*T, ...) in frame-major order
                feat = process_volumes(v, mode="slice", ijk=ijk_t.reshape(B, T, 3), proj_mask=proj_mask, scale=scale)
                ovo, ovr, vote, proba, lab = self._decide(feat, want_proba=want_proba)
                out = {"dec_ovo": ovo, "dec_ovr": ovr, "label_vote": vote}
                if want_proba:
                    out["proba"], out["label_calib"] = proba, lab
                return out
        out = {
            "dec_ovo": torch.empty((B, P), dtype=torch.float64, device=dev),
            "dec_ovr": torch.empty((B,) if C_ == 2 else (B, C_), dtype=torch.float64, device=dev),
            "label_vote": torch.empty((B,), dtype=torch.int32, device=dev),
        }
        if want_proba:
            out["proba"] = torch.empty((B, C_), dtype=torch.float64, device=dev)
            out["label_calib"] = torch.empty((B,), dtype=torch.int32, device=dev)
        if fused_derive:
            out["ijk"] = torch.empty((B, 3), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.rml_derive_project_svm(
                    self._ctx, self._h, _lib.ptr(v), vdt, B, X, Y, Z, float(RADAR_MAX) if scale else 0.0, _mask_bits(proj_mask),
                    _lib.ptr(out["ijk"]), _lib.ptr(out["dec_ovo"]), _lib.ptr(out["dec_ovr"]), _lib.ptr(out.get("proba")),
                    _lib.ptr(out["label_vote"]), _lib.ptr(out.get("label_calib")), _lib.stream_ptr(dev)), "rml_derive_project_svm")
            return out
        with torch.cuda.device(dev):
            _lib.check(lib.rml_project_svm(
                self._ctx, self._h, _lib.ptr(v), vdt, B, X, Y, Z, _lib.MODES[mode], _lib.ptr(ijk_t),
                float(RADAR_MAX) if scale else 0.0, _mask_bits(proj_mask),
                _lib.ptr(out["dec_ovo"]), _lib.ptr(out["dec_ovr"]), _lib.ptr(out.get("proba")),
                _lib.ptr(out["label_vote"]), _lib.ptr(out.get("label_calib")), _lib.stream_ptr(dev)), "rml_project_svm")
        return out

    def kernel_matrix(self, X, path=None):
        """K[n][m] = k(x_n, sv_m) as a float64 CUDA tensor (N, n_SV): the kernel evaluations of ``decision_function``
        themselves (exact-integer path on code-grid rows, float64 MFMA elsewhere)."""
        torch = _torch()
        lib = _lib.load()
        Xd = self._rows(X)
        N = Xd.shape[0]
        K = torch.empty((N, self.n_sv), dtype=torch.float64, device=Xd.device)
        with torch.cuda.device(Xd.device):
            _lib.check(lib.rml_svm_kernel_matrix(self._ctx, self._h, _lib.PATHS[path or self.path], _lib.ptr(Xd), Xd.stride(0), N,
                                                 _lib.ptr(K), K.stride(0), _lib.stream_ptr(Xd.device)), "rml_svm_kernel_matrix")
        return K

    # -- sklearn protocol -------------------------------------------------------------------
    def decision_function(self, X):
        """SVC.decision_function (sk:svm/_base.py:760-790): (N,C) 'ovr' scores (or the raw
        (N,P) libsvm pair values for decision_function_shape='ovo'; (N,) for 2 classes)."""
        ovo, ovr, _, _, _ = self._decide(self._rows(X))
        if len(self.classes_) > 2 and self.decision_function_shape == "ovo":
            return ovo.cpu().numpy()
        return ovr.cpu().numpy()

    def predict_proba(self, X):
        """SVC.predict_proba of a probability=True model: libsvm's Platt sigmoids + pairwise coupling
        (sk:svm/src/libsvm/svm.cpp:2918-2952).  (The reference itself calls predict_proba on the calibrated
        wrapper -- GpuCalibratedClassifier -- not on the bare SVC.)"""
        if self._probA is None:
            raise AttributeError("predict_proba is not available when probability=False")
        torch = _torch()
        lib = _lib.load()
        Xd = self._rows(X)
        ovo = self._decide(Xd)[0]
        N, C_ = Xd.shape[0], len(self.classes_)
        proba = torch.empty((N, C_), dtype=torch.float64, device=Xd.device)
        with torch.cuda.device(Xd.device):
            _lib.check(lib.rml_svm_pairwise_proba(self._ctx, self._h, _lib.ptr(ovo), N, _lib.ptr(proba),
                                                  _lib.stream_ptr(Xd.device)), "rml_svm_pairwise_proba")
        return proba.cpu().numpy()

    def predict(self, X):
        """SVC.predict: libsvm one-vs-one vote (sk:svm/src/libsvm/svm.cpp:2884-2894)."""
        _, _, vote, _, _ = self._decide(self._rows(X))
        return self.classes_.take(vote.cpu().numpy().astype(np.intp))


class GpuCalibratedClassifier(_Base):
    """GPU twin of ``CalibratedClassifierCV(estimator, cv='prefit')`` with sigmoid calibrators
    (the object the reference pickles: train.py:722-731).  Wraps a GpuSVC or GpuLinearClassifier."""

    def __init__(self, estimator):
        self.estimator = estimator
        self.classes_ = estimator.classes_
        self.n_features_in_ = estimator.n_features_in_
        self._dev = estimator._dev

    def predict_proba(self, X):
        """sk:calibration.py:492-518,727-784."""
        return self.estimator._proba(self._rows(X))[0].cpu().numpy()

    def predict(self, X):
        """sk:calibration.py:520-537: classes_[argmax(predict_proba)]."""
        lab = self.estimator._proba(self._rows(X))[1]
        return self.classes_.take(lab.cpu().numpy().astype(np.intp))

    def decision_function(self, X):
        return self.estimator.decision_function(X)


def _svc_proba(self, Xd):
    _, _, _, proba, lab = self._decide(Xd, want_proba=True)
    return proba, lab


GpuSVC._proba = _svc_proba


class GpuLinearClassifier(_Base):
    """GPU twin of a fitted ``SGDClassifier`` (train.py:350-381): decision = X @ coef_.T + intercept_,
    predict = argmax (train.py:421,433)."""

    def __init__(self, coef, intercept, classes, calib_a=None, calib_b=None, device=None):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.RadarMLError("no HIP device is visible: the radar-ml HIP path needs an MI355X (no CPU fallback)")
        lib = _lib.load()
        cf, ic = _f64(coef), _f64(intercept)
        self.classes_ = np.asarray(classes)
        self.n_features_in_ = cf.shape[1]
        C_ = len(self.classes_)
        _check_classes(C_)
        self._dev = _lib.device_of(device)
        self._ctx = _lib.context(self._dev)
        ca = cb = None
        if calib_a is not None:
            ca, cb = _f64(calib_a), _f64(calib_b)
        h = C.c_void_p()
        with torch.cuda.device(self._dev):
            _lib.check(lib.rml_linear_load(self._ctx, cf.ctypes.data, ic.ctypes.data, C_, cf.shape[1],
                                           ca.ctypes.data if ca is not None else None,
                                           cb.ctypes.data if cb is not None else None, C.byref(h)), "rml_linear_load")
        self._h = h
        self.has_calibration = ca is not None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().rml_linear_free(self._ctx, self._h)
                self._h = None
        except Exception:
            pass

    def _run(self, Xd, want_proba=False):
        torch = _torch()
        lib = _lib.load()
        N = Xd.shape[0]
        C_ = len(self.classes_)
        dev = Xd.device
        dec = torch.empty((N,) if C_ == 2 else (N, C_), dtype=torch.float64, device=dev)
        lab = torch.empty((N,), dtype=torch.int32, device=dev)
        proba = labc = None
        if want_proba:
            proba = torch.empty((N, C_), dtype=torch.float64, device=dev)
            labc = torch.empty((N,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.rml_linear_decision(self._ctx, self._h, _lib.ptr(Xd), Xd.stride(0), N, _lib.ptr(dec), _lib.ptr(proba),
                                               _lib.ptr(lab), _lib.ptr(labc), _lib.stream_ptr(dev)), "rml_linear_decision")
        return dec, lab, proba, labc

    def decision_function(self, X):
        return self._run(self._rows(X))[0].cpu().numpy()

    def predict(self, X):
        lab = self._run(self._rows(X))[1]
        return self.classes_.take(lab.cpu().numpy().astype(np.intp))

    def decide_volumes(self, volumes, mode="max", ijk=None, proj_mask=ProjMask(True, True, True), scale=True):
        """Batched path for the reference's default (SGD) model: (B,X,Y,Z) volumes -> projection + feature rows
        (one pass over the volumes) -> linear scores, labels and calibrated probabilities, all on the GPU."""
        from .common import process_volumes
        feat = process_volumes(volumes, mode=mode, ijk=ijk, proj_mask=proj_mask, scale=scale, device=self._dev)
        dec, lab, proba, labc = self._run(feat, want_proba=self.has_calibration)
        out = {"dec": dec, "label": lab}
        if proba is not None:
            out["proba"] = proba
            out["label_calib"] = labc
        return out

    def _proba(self, Xd):
        _, _, proba, labc = self._run(Xd, want_proba=True)
        return proba, labc


class KernelMatrix:
    """Gram-matrix service for fitting ``SVC(kernel='precomputed')`` on the GPU-resident kernel (the reference's RBF
    grid search, train.py:462-491, spends its time in libsvm's pairwise kernel evaluations): the training rows are
    loaded once, ``gram()`` is K(train, train), ``against(X)`` is K(X, train) for validation / test rows."""

    def __init__(self, X_train, gamma, kernel="rbf", device=None):
        Xt = _f64(X_train)
        M = Xt.shape[0]
        if M < 2:
            raise ValueError("KernelMatrix: at least two training rows")
        self._svc = GpuSVC(Xt, np.zeros((1, M)), np.zeros(1), np.array([M - 1, 1], dtype=np.int32), gamma,
                           np.arange(2), kernel=kernel, device=device)
        self._train = Xt
        self.exact = self._svc.exact

    def gram(self):
        return self._svc.kernel_matrix(self._train)

    def against(self, X):
        return self._svc.kernel_matrix(X)


def unwrap_calibrated(cc):
    """(estimator, a, b) of one ``_CalibratedClassifier``.  scikit-learn >= 1.2 names the parts ``estimator`` /
    ``calibrators``; the release the reference pins (0.24.0, requirements.txt:57 -- what its own pickles were written
    with, train.py:722-731) names them ``base_estimator`` / ``calibrators_``.  Pure attribute access: no GPU needed."""
    if getattr(cc, "method", "sigmoid") != "sigmoid":
        raise NotImplementedError("only sigmoid calibration (the sklearn default the reference uses)")
    est = getattr(cc, "estimator", None)
    if est is None:
        est = getattr(cc, "base_estimator", None)
    cals = getattr(cc, "calibrators", None)
    if cals is None:
        cals = getattr(cc, "calibrators_", None)
    if est is None or cals is None:
        raise NotImplementedError("unrecognised _CalibratedClassifier layout: %s" % sorted(vars(cc)))
    if type(est).__name__ == "FrozenEstimator":
        est = est.estimator
    a = np.array([c.a_ for c in cals], dtype=np.float64)
    b = np.array([c.b_ for c in cals], dtype=np.float64)
    return est, a, b


def from_sklearn(obj, **kw):
    """Ingest the fitted object the reference pickles (train.py:729-731) or a bare estimator.

    CalibratedClassifierCV(prefit, sigmoid) -> GpuCalibratedClassifier; SVC -> GpuSVC;
    SGDClassifier -> GpuLinearClassifier."""
    name = type(obj).__name__
    if name == "CalibratedClassifierCV":
        ccs = obj.calibrated_classifiers_
        if len(ccs) != 1:
            raise NotImplementedError("only cv='prefit' calibration (one calibrated classifier) is used by the reference")
        est, a, b = unwrap_calibrated(ccs[0])
        if type(est).__name__ == "SVC":
            return GpuCalibratedClassifier(GpuSVC.from_sklearn(est, calib=(a, b), **kw))
        if type(est).__name__ == "SGDClassifier":
            return GpuCalibratedClassifier(GpuLinearClassifier(est.coef_, est.intercept_, est.classes_, a, b, **kw))
        raise NotImplementedError("calibrated %s" % type(est).__name__)
    if name == "SVC":
        return GpuSVC.from_sklearn(obj, **kw)
    if name == "SGDClassifier":
        return GpuLinearClassifier(obj.coef_, obj.intercept_, obj.classes_, **kw)
    raise NotImplementedError("cannot ingest %s" % name)
