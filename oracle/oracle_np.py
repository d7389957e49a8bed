"""CPU oracle for the radar-ml projection -> feature -> classifier hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker.  The product path (``radar-ml_amd/``) never
imports this module and fails loudly when the HIP library is missing.

This is a NumPy restatement (float64 where the reference computes in float64) of the
algorithms on the hot path of goruck/radar-ml.  Every function cites the reference
``file:line`` it follows (paths relative to /root/reference) or, for the SVM
arithmetic that lives in the reference's third-party dependency scikit-learn
(pinned ``scikit-learn==0.24.0`` in requirements.txt:57; the container has 1.7.2,
same libsvm algorithm), the ``sk:`` path inside the installed sklearn package.

Parity pinning: the reference has no tests of its own.  This
oracle is pinned by
  * the 28 (x,y,z)->(i,j,k) known answers printed in ground_truth_samples.log,
  * the reference's ``common.py`` imported in the build container
    (tests/golden/make_golden.py, stub WalabotAPI), and
  * scikit-learn 1.7.2 (``SVC`` / ``CalibratedClassifierCV`` / ``SGDClassifier``)
    imported in the build container,
all captured as committed fixtures under tests/golden/.
"""
from __future__ import annotations

import collections

import numpy as np

# --------------------------------------------------------------------------------------
# Arena constants -- common.py:25-31
# --------------------------------------------------------------------------------------
R_MIN, R_MAX, R_RES = 10, 360, 2
THETA_MIN, THETA_MAX, THETA_RES = -42, 42, 4
PHI_MIN, PHI_MAX, PHI_RES = -30, 30, 2
RADAR_MIN = 0.0
RADAR_MAX = 255.0

# common.py:40,43 -- tuple order is (xz, yz, xy)
ProjMask = collections.namedtuple("ProjMask", ["xz", "yz", "xy"])
ProjZoom = collections.namedtuple("ProjZoom", ["xz", "yz", "xy"])


# --------------------------------------------------------------------------------------
# Geometry -- common.py:93-121
# --------------------------------------------------------------------------------------
def cartesian_to_spherical(x, y, z):
    """common.py:93-97."""
    r = np.sqrt(np.power(x, 2) + np.power(y, 2) + np.power(z, 2))
    phi = np.arctan2(y, z)
    theta = np.arcsin(x / r)
    return (r, np.rad2deg(theta), np.rad2deg(phi))


def spherical_to_cartesian(r, theta, phi):
    """common.py:99-104."""
    theta_rad, phi_rad = np.deg2rad(theta), np.deg2rad(phi)
    x = r * np.sin(theta_rad)
    y = r * np.cos(theta_rad) * np.sin(phi_rad)
    z = r * np.cos(theta_rad) * np.cos(phi_rad)
    return (x, y, z)


def calculate_matrix_indices(x, y, z, size_x, size_y, size_z):
    """common.py:106-121.  Truncation toward zero (Python int()), no clamping."""
    r, theta, phi = cartesian_to_spherical(x, y, z)
    i = int((theta - THETA_MIN) * (size_x - 1) / (THETA_MAX - THETA_MIN))
    j = int((phi - PHI_MIN) * (size_y - 1) / (PHI_MAX - PHI_MIN))
    k = int((r - R_MIN) * (size_z - 1) / (R_MAX - R_MIN))
    return (i, j, k)


# --------------------------------------------------------------------------------------
# 3-D -> 2-D projections
# --------------------------------------------------------------------------------------
def project_slice(volume, i, j, k):
    """Plane slices through the target voxel, tuple order (xz, yz, xy).

    ground_truth_samples.py:413-419,440 and predict.py:102-107,113:
    ``yz = V[i,:,:]``, ``xz = V[:,j,:]``, ``xy = V[:,:,k]``.  Python negative-index
    semantics apply (calculate_matrix_indices does not clamp).
    """
    v = np.asarray(volume)
    return (v[:, j, :], v[i, :, :], v[:, :, k])


def project_max(volume):
    """Max-projection form named by BASELINE.json (SURVEY.md §0.1 D1, §8 a-1').

    ``xz = max_j V``, ``yz = max_i V``, ``xy = max_k V``; same tuple order/layout as
    :func:`project_slice`.  Exact and order independent, hence bit-exact on any
    implementation.  Accepts (X,Y,Z) or batched (B,X,Y,Z).
    """
    v = np.asarray(volume)
    if v.ndim == 3:
        return (v.max(axis=1), v.max(axis=0), v.max(axis=2))
    return (v.max(axis=2), v.max(axis=1), v.max(axis=3))


def project_sum(volume):
    """Sum-reductions used by DerivedTarget (common.py:51-53), planes in (xz,yz,xy) order."""
    v = np.asarray(volume)
    if v.ndim == 3:
        return (v.sum(axis=1), v.sum(axis=0), v.sum(axis=2))
    return (v.sum(axis=2), v.sum(axis=1), v.sum(axis=3))


def features_from_projections(xz, yz, xy, proj_mask=(True, True, True), scale=False):
    """Batched feature assembly at zoom == 1 (identity zoom): common.py:141-148.

    xz (B,X,Z), yz (B,Y,Z), xy (B,X,Y) -> (B,D) float32, concatenation order
    xz | yz | xy, C-order ravel, optional ``/ RADAR_MAX`` carried out in float32
    exactly as ``concat_projections / RADAR_MAX`` does on a float32 array.
    """
    parts = []
    for keep, p in zip(proj_mask, (xz, yz, xy)):
        if keep:
            p = np.asarray(p, dtype=np.float32)
            parts.append(p.reshape(p.shape[0], -1))
    f = np.concatenate(parts, axis=1)
    if scale:
        f = f / np.float32(RADAR_MAX)
    return np.ascontiguousarray(f, dtype=np.float32)


# --------------------------------------------------------------------------------------
# Derived targets -- common.py:45-80
# --------------------------------------------------------------------------------------
DerivedTarget = collections.namedtuple(
    "DerivedTarget", ["xPosCm", "yPosCm", "zPosCm", "amplitude", "i", "j", "k"]
)


def axis_energy_profiles(volume):
    """The three 1-D energy profiles of common.py:51-53 (find_max_indices' ``sums``).

    ``find_max_indices(1,1)`` -> sum over axis 1 then axis 1 of the result = profile
    over i (theta); ``(0,1)`` -> profile over j (phi); ``(0,0)`` -> profile over k (r).
    """
    v = np.asarray(volume)
    s_theta = np.sum(np.sum(v, axis=1), axis=1)
    s_phi = np.sum(np.sum(v, axis=0), axis=1)
    s_r = np.sum(np.sum(v, axis=0), axis=0)
    return s_theta, s_phi, s_r


def find_max_indices(sums, num_targets=1):
    """common.py:54-55: top-``num_targets`` indices ascending by value."""
    max_indices = np.argpartition(sums, -num_targets)[-num_targets:]
    return max_indices[np.argsort(sums[max_indices])]


def get_derived_targets(radar_data, size_x, size_y, size_z, num_targets=1):
    """common.py:49-80."""
    s_theta, s_phi, s_r = axis_energy_profiles(radar_data)
    it = find_max_indices(s_theta, num_targets)
    ip = find_max_indices(s_phi, num_targets)
    ir = find_max_indices(s_r, num_targets)

    def make(i, j, k):
        theta = THETA_MIN + i * (THETA_MAX - THETA_MIN) / (size_x - 1)
        phi = PHI_MIN + j * (PHI_MAX - PHI_MIN) / (size_y - 1)
        r = R_MIN + k * (R_MAX - R_MIN) / (size_z - 1)
        x, y, z = spherical_to_cartesian(r, theta, phi)
        return DerivedTarget(x, y, z, None, i, j, k)

    return [make(i, j, k) for i, j, k in zip(it, ip, ir)]


# --------------------------------------------------------------------------------------
# process_samples -- common.py:123-149
# --------------------------------------------------------------------------------------
def process_samples(samples, proj_mask=ProjMask(xz=True, yz=True, xy=True),
                    proj_zoom=ProjZoom(xz=[1.0, 1.0], yz=[1.0, 1.0], xy=[1.0, 1.0]),
                    scale=False):
    """common.py:123-149, restated with the same SciPy call (ndimage.zoom defaults:
    order-3 spline, mode='constant', prefilter=True)."""
    from scipy import ndimage

    def make(t):
        wanted = tuple(ndimage.zoom(p, proj_zoom[i]) for i, p in enumerate(t) if proj_mask[i])
        concat = np.concatenate(wanted, axis=None)
        return concat / RADAR_MAX if scale else concat

    return np.array([make(t) for t in samples])


def calc_proj_zoom(train_size_x, train_size_y, train_size_z, size_x, size_y, size_z):
    """predict.py:34-54."""
    x_zoom = train_size_x / size_x
    y_zoom = train_size_y / size_y
    z_zoom = train_size_z / size_z
    return ProjZoom(xy=[x_zoom, y_zoom], xz=[x_zoom, z_zoom], yz=[y_zoom, z_zoom])


# --------------------------------------------------------------------------------------
# SVM decision function -- sk:svm/src/libsvm/svm.cpp
# --------------------------------------------------------------------------------------
def svm_kernel_values(X, SV, gamma, kernel="rbf", block=64):
    """K[n,m] as libsvm computes it: sk:svm/src/libsvm/svm.cpp:461-475,514.

    RBF: float64 direct difference ``m = x - sv; sum = dot(m, m); exp(-gamma*sum)``
    (no norm expansion).  LINEAR: ``dot(x, sv)`` (svm.cpp:457).  X is first cast to
    float64 C-order as sk:svm/_base.py:610-620 does.
    """
    X = np.ascontiguousarray(X, dtype=np.float64)
    SV = np.ascontiguousarray(SV, dtype=np.float64)
    N, M = X.shape[0], SV.shape[0]
    K = np.empty((N, M), dtype=np.float64)
    if kernel == "linear":
        np.dot(X, SV.T, out=K)
        return K
    for n0 in range(0, N, block):
        xb = X[n0:n0 + block]
        for m in range(M):
            d = xb - SV[m]
            K[n0:n0 + block, m] = np.einsum("nd,nd->n", d, d)
    return np.exp(-gamma * K)


def ovo_weight_matrix(dual_coef, n_support):
    """Per-pair SV weights W (P,M) so that dec[:,p] = K @ W[p] + intercept[p].

    Restates the pair loop of svm_predict_values, sk:svm/src/libsvm/svm.cpp:2864-2890:
    for pair p=(i<j): coef1 = sv_coef[j-1] over class-i SVs, coef2 = sv_coef[i] over
    class-j SVs.
    """
    dual_coef = np.asarray(dual_coef, dtype=np.float64)
    n_support = np.asarray(n_support, dtype=np.int64)
    C = n_support.shape[0]
    M = dual_coef.shape[1]
    start = np.concatenate([[0], np.cumsum(n_support)[:-1]])
    P = C * (C - 1) // 2
    W = np.zeros((P, M), dtype=np.float64)
    p = 0
    for i in range(C):
        for j in range(i + 1, C):
            si, sj, ci, cj = start[i], start[j], n_support[i], n_support[j]
            W[p, si:si + ci] = dual_coef[j - 1, si:si + ci]
            W[p, sj:sj + cj] = dual_coef[i, sj:sj + cj]
            p += 1
    return W


def svm_decision_ovo(X, SV, dual_coef, intercept, n_support, gamma, kernel="rbf"):
    """libsvm ``dec_values`` (N,P): sk:svm/src/libsvm/svm.cpp:2847-2890.

    ``rho[p] = -intercept_[p]`` (sk:svm/src/libsvm/libsvm_helper.c:168-170), so
    ``dec = sum(coef*K) + intercept``.  Summation order follows libsvm (class-i block
    then class-j block, sequential float64), reproduced here per pair with
    sequential accumulation on the non-zero weights.
    """
    K = svm_kernel_values(X, SV, gamma, kernel)
    n_support = np.asarray(n_support, dtype=np.int64)
    C = n_support.shape[0]
    start = np.concatenate([[0], np.cumsum(n_support)[:-1]])
    dual_coef = np.asarray(dual_coef, dtype=np.float64)
    intercept = np.asarray(intercept, dtype=np.float64)
    N = K.shape[0]
    P = C * (C - 1) // 2
    dec = np.zeros((N, P), dtype=np.float64)
    p = 0
    for i in range(C):
        for j in range(i + 1, C):
            si, sj, ci, cj = start[i], start[j], n_support[i], n_support[j]
            # sequential float64 accumulation in SV order, as the C loops do
            s = np.zeros(N, dtype=np.float64)
            for k in range(ci):
                s += dual_coef[j - 1, si + k] * K[:, si + k]
            for k in range(cj):
                s += dual_coef[i, sj + k] * K[:, sj + k]
            dec[:, p] = s + intercept[p]   # sum -= rho[p]
            p += 1
    return dec


def svm_vote_labels(dec, n_classes):
    """OvO vote of svm_predict_values: sk:svm/src/libsvm/svm.cpp:2884-2894.

    ``dec>0 -> ++vote[i] else ++vote[j]``; first maximum wins.  Returns class
    *indices* (position in ``classes_``)."""
    dec = np.asarray(dec)
    N = dec.shape[0]
    votes = np.zeros((N, n_classes), dtype=np.int64)
    p = 0
    for i in range(n_classes):
        for j in range(i + 1, n_classes):
            pos = dec[:, p] > 0
            votes[pos, i] += 1
            votes[~pos, j] += 1
            p += 1
    return np.argmax(votes, axis=1)   # first max


def ovr_decision_function(dec, n_classes):
    """sk:svm/_base.py:780-790 + sk:utils/multiclass.py:542-584.

    ``_ovr_decision_function(dec < 0, -dec, n_classes)``: votes + s/(3(|s|+1))."""
    dec = np.asarray(dec, dtype=np.float64)
    predictions = dec < 0
    confidences = -dec
    N = dec.shape[0]
    votes = np.zeros((N, n_classes))
    soc = np.zeros((N, n_classes))
    k = 0
    for i in range(n_classes):
        for j in range(i + 1, n_classes):
            soc[:, i] -= confidences[:, k]
            soc[:, j] += confidences[:, k]
            votes[predictions[:, k] == 0, i] += 1
            votes[predictions[:, k] == 1, j] += 1
            k += 1
    return votes + soc / (3 * (np.abs(soc) + 1))


def sklearn_decision_function(dec, n_classes):
    """What SVC.decision_function returns (sk:svm/_base.py:546-547,780-790): binary problems flip the sign of
    the single libsvm pair value and return shape (N,); otherwise the 'ovr' transform."""
    dec = np.asarray(dec, dtype=np.float64)
    if n_classes == 2:
        return -dec[:, 0]
    return ovr_decision_function(dec, n_classes)


def expit(x):
    """scipy.special.expit restated: 1/(1+exp(-x)), overflow-safe."""
    x = np.asarray(x, dtype=np.float64)
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    e = np.exp(x[~pos])
    out[~pos] = e / (1.0 + e)
    return out


def calibrated_proba(T, calib_a, calib_b):
    """_CalibratedClassifier.predict_proba for the sigmoid method, >2 classes.

    sk:calibration.py:727-784 (normalise, uniform when the row sums to 0, clip
    (1, 1+1e-5] -> 1) and sk:calibration.py:928-942 (``expit(-(a*T+b))``).
    ``T`` is the estimator's decision_function output (N,C)."""
    T = np.asarray(T, dtype=np.float64)
    a = np.asarray(calib_a, dtype=np.float64)
    b = np.asarray(calib_b, dtype=np.float64)
    if T.ndim == 1:      # binary: one calibrator for classes_[1]; proba[:,0] = 1 - proba[:,1] (calibration.py:760-767)
        p1 = expit(-(a[0] * T + b[0]))
        proba = np.stack([1.0 - p1, p1], axis=1)
        proba[(1.0 < proba) & (proba <= 1.0 + 1e-5)] = 1.0
        return proba
    n_classes = T.shape[1]
    proba = expit(-(a[None, :] * T + b[None, :]))
    den = proba.sum(axis=1)[:, None]
    uniform = np.full_like(proba, 1.0 / n_classes)
    proba = np.divide(proba, den, out=uniform, where=den != 0)
    proba[(1.0 < proba) & (proba <= 1.0 + 1e-5)] = 1.0
    return proba


def calibrated_labels(proba):
    """CalibratedClassifierCV.predict: sk:calibration.py:520-537 -> argmax (first max)."""
    return np.argmax(proba, axis=1)


def libsvm_pairwise_proba(dec, probA, probB, n_classes):
    """SVC.predict_proba for probability=True models (the SVC of train.py:478): libsvm's
    svm_predict_probability, sk:svm/src/libsvm/svm.cpp:2032-2040 (sigmoid_predict), 2043-2104
    (multiclass_probability, method 2 of Wu, Lin & Weng), 2918-2952.  ``dec`` are libsvm's pair values (N,P)."""
    dec = np.asarray(dec, dtype=np.float64).reshape(len(dec), -1)
    k = n_classes
    N = dec.shape[0]
    out = np.empty((N, k))
    for n in range(N):
        r = np.zeros((k, k))
        q = 0
        for i in range(k):
            for j in range(i + 1, k):
                f = dec[n, q] * probA[q] + probB[q]
                s = np.exp(-f) / (1.0 + np.exp(-f)) if f >= 0 else 1.0 / (1.0 + np.exp(f))
                r[i, j] = min(max(s, 1e-7), 1 - 1e-7)
                r[j, i] = 1 - r[i, j]
                q += 1
        p = np.full(k, 1.0 / k)
        Q = np.zeros((k, k))
        for t in range(k):
            for j in range(t):
                Q[t, t] += r[j, t] * r[j, t]
                Q[t, j] = Q[j, t]
            for j in range(t + 1, k):
                Q[t, t] += r[j, t] * r[j, t]
                Q[t, j] = -r[j, t] * r[t, j]
        eps = 0.005 / k
        for it in range(max(100, k)):
            Qp = Q @ p
            pQp = float(p @ Qp)
            if np.abs(Qp - pQp).max() < eps:
                break
            for t in range(k):
                diff = (-Qp[t] + pQp) / Q[t, t]
                p[t] += diff
                pQp = (pQp + diff * (diff * Q[t, t] + 2 * Qp[t])) / (1 + diff) / (1 + diff)
                Qp = (Qp + diff * Q[t]) / (1 + diff)
                p = p / (1 + diff)
        out[n] = p
    return out


def linear_decision(X, coef, intercept):
    """SGDClassifier.decision_function: X @ coef_.T + intercept_ (train.py:421,433 predict
    = argmax of this for >2 classes; sk:linear_model/_base.py decision_function)."""
    X = np.asarray(X, dtype=np.float64)
    return X @ np.asarray(coef, dtype=np.float64).T + np.asarray(intercept, dtype=np.float64)


def classifier_threshold(proba, class_names, min_proba=0.7):
    """predict.py:56-70, batched: (name, proba) per row; 'Unknown' below min_proba."""
    proba = np.asarray(proba)
    j = np.argmax(proba, axis=1)
    p = proba[np.arange(proba.shape[0]), j]
    names = [class_names[jj] if pp >= min_proba else "Unknown" for jj, pp in zip(j, p)]
    return names, p


# --------------------------------------------------------------------------------------
# Synthetic radar volumes for tests / the CPU baseline.  The device
# generator (csrc/synth.hip) draws from the same family of frames but is NOT required to
# match this one bit-for-bit: parity always runs the oracle on the very same volumes the
# HIP path consumed (tests upload these; bench.py downloads a slab of device frames).
# --------------------------------------------------------------------------------------
def synth_volumes(seed, nframes, X, Y, Z, n_classes=3, dtype=np.float32):
    """Integer-valued sparse Gaussian-blob radar returns (nframes,X,Y,Z) + class ids.

    Model of the real data: background exactly 0, 1-3 separable
    Gaussian blobs per frame, peak amplitude U[76,255], values rounded to integers,
    values below 13 set to 0, clipped to 255; blob size grows with the class index so
    a classifier has signal.
    """
    rng = np.random.default_rng(seed)
    cls = rng.integers(0, n_classes, size=nframes).astype(np.int32)
    vol = np.zeros((nframes, X, Y, Z), dtype=np.float32)
    ix = np.arange(X, dtype=np.float32)[None, :]
    iy = np.arange(Y, dtype=np.float32)[None, :]
    iz = np.arange(Z, dtype=np.float32)[None, :]
    k = cls.astype(np.float32) + 1.0
    nblob = 1 + (cls % 3)
    for b in range(3):
        cx = rng.random(nframes, dtype=np.float32) * (X - 1)
        cy = rng.random(nframes, dtype=np.float32) * (Y - 1)
        cz = rng.random(nframes, dtype=np.float32) * (Z - 1)
        sx = (1.0 + rng.random(nframes, dtype=np.float32) * 1.5) * (0.6 + 0.4 * k)
        sy = (1.0 + rng.random(nframes, dtype=np.float32) * 1.5) * (0.6 + 0.4 * k)
        sz = (3.0 + rng.random(nframes, dtype=np.float32) * 7.0) * (0.6 + 0.4 * k)
        amp = np.floor(76.0 + rng.random(nframes, dtype=np.float32) * 179.0) * (b < nblob)
        gx = np.exp(-0.5 * ((ix - cx[:, None]) / sx[:, None]) ** 2).astype(np.float32)
        gy = np.exp(-0.5 * ((iy - cy[:, None]) / sy[:, None]) ** 2).astype(np.float32)
        gz = np.exp(-0.5 * ((iz - cz[:, None]) / sz[:, None]) ** 2).astype(np.float32)
        v = (amp[:, None, None, None] * gx[:, :, None, None]) * gy[:, None, :, None] * gz[:, None, None, :]
        np.maximum(vol, v, out=vol)
    vol = np.floor(vol + np.float32(0.5))
    np.minimum(vol, np.float32(255.0), out=vol)
    vol[vol < np.float32(13.0)] = 0.0
    return vol.astype(dtype), cls


# ----------------------------------------------------------------------------------------------------------------
# PIL bicubic resize of float32 ('F' mode) images -- the resize in front of the dnn / sgan classifiers
# (dnn.py:240-245, sgan.py:676-681: Image.fromarray(p).resize(RESCALE, resample=Image.BICUBIC)).
# The algorithm lives in Pillow (requirements.txt:41 pins Pillow 8.2.0; absent from /root/reference), file
# src/libImaging/Resample.c: precompute_coeffs() + ImagingResampleHorizontal_32bpc / ImagingResampleVertical_32bpc.
# Restated here; pinned against Pillow itself by tests/golden/pil_resize.npz (tests/golden/make_golden.py).
# ----------------------------------------------------------------------------------------------------------------
def _pil_bicubic_filter(x):
    a = -0.5
    x = -x if x < 0.0 else x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_resample_coeffs(in_size, out_size):
    """precompute_coeffs() of Resample.c for the full box and the bicubic filter (support 2): per output index the
    first input index, the tap count and the normalised double weights.  Returns (bounds (out,2) int32, kk (out,ksize) f64)."""
    scale = float(in_size) / float(out_size)
    filterscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)         # C truncation of a value >= -support + 0.5 ...
        if center - support + 0.5 < 0:
            xmin = 0                               # ... and the clamp at 0 (truncation towards zero never goes below)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        ww = 0.0
        for x in range(xmax):
            w = _pil_bicubic_filter((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            for x in range(xmax):
                kk[xx, x] /= ww
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def pil_resize_bicubic(img, out_hw):
    """Image.fromarray(img).resize((out_w, out_h), Image.BICUBIC) for a 2-D float array: horizontal pass into a
    float32 intermediate, then the vertical pass; every output is a double-precision sum in tap order, rounded to
    float32 (no fused multiply-add).  A pass whose size does not change is skipped, as in ImagingResample()."""
    src = np.asarray(img, dtype=np.float32)
    H, W = src.shape
    OH, OW = out_hw
    cur = src
    if OW != W:
        bounds, kk = pil_resample_coeffs(W, OW)
        tmp = np.empty((H, OW), np.float32)
        for xx in range(OW):
            x0, n = bounds[xx]
            ss = np.zeros(H, np.float64)
            for t in range(n):
                ss = ss + cur[:, x0 + t].astype(np.float64) * kk[xx, t]
            tmp[:, xx] = ss.astype(np.float32)
        cur = tmp
    if OH != H:
        bounds, kk = pil_resample_coeffs(H, OH)
        out = np.empty((OH, cur.shape[1]), np.float32)
        for yy in range(OH):
            y0, n = bounds[yy]
            ss = np.zeros(cur.shape[1], np.float64)
            for t in range(n):
                ss = ss + cur[y0 + t, :].astype(np.float64) * kk[yy, t]
            out[yy, :] = ss.astype(np.float32)
        cur = out
    return np.array(cur, dtype=np.float32, copy=True)


def scale_unit_range(p, radar_max=255.0):
    """dnn.py:202-205 / sgan.py:638-641: (p - RADAR_MAX/2.) / (RADAR_MAX/2.) with NumPy's dtype rules -- a float32
    projection stays float32 through both operations (Python scalars do not widen it), anything else goes through
    float64 -- then Image.fromarray stores float32 ('F' mode)."""
    p = np.asarray(p)
    if p.dtype == np.float32:
        h = np.float32(radar_max / 2.0)
        return ((p - h) / h).astype(np.float32)
    return ((p.astype(np.float64) - radar_max / 2.0) / (radar_max / 2.0)).astype(np.float32)


# ----------------------------------------------------------------------------------------------------------------
# Multi-view CNN forward (dnn.py:45-91), Keras semantics restated in NumPy float64.  "Parity unpinned": the
# reference ships no weights and TensorFlow is not installable here, so this pins the layer semantics (TF 'same'
# padding, NHWC flatten order, branch order) for the PyTorch module and the HIP trunk, not trained outputs.
# Weights in Keras layout: conv kernels (kh, kw, cin, cout), dense kernels (in, out).
# ----------------------------------------------------------------------------------------------------------------
def keras_conv2d_same_s2_relu(x, kernel, bias):
    """Conv2D(filters, (3,3), strides=(2,2), padding='same', activation='relu') on NHWC input (dnn.py:47-50).
    TF 'same': out = ceil(in/2), total pad = max((out-1)*2 + 3 - in, 0), pad_before = total // 2 (the extra row /
    column goes to the bottom / right)."""
    x = np.asarray(x, np.float64)
    n, h, w, cin = x.shape
    kh, kw, _, cout = kernel.shape
    oh, ow = -(-h // 2), -(-w // 2)
    ph = max((oh - 1) * 2 + kh - h, 0)
    pw = max((ow - 1) * 2 + kw - w, 0)
    xp = np.zeros((n, h + ph, w + pw, cin))
    xp[:, ph // 2:ph // 2 + h, pw // 2:pw // 2 + w, :] = x
    out = np.zeros((n, oh, ow, cout))
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky:ky + 2 * oh:2, kx:kx + 2 * ow:2, :]          # (n, oh, ow, cin)
            out += patch @ np.asarray(kernel[ky, kx], np.float64)
    return np.maximum(out + np.asarray(bias, np.float64), 0.0)


def dnn_conv_features(xz, yz, xy, conv_weights):
    """create_conv_layers on each input, Concatenate()([xz, yz, xy]) on the channel axis, Flatten (dnn.py:68-76):
    (N, H/4 * W/4 * 96) rows in (h, w, c) order.  conv_weights: per branch (k1, b1, k2, b2)."""
    outs = []
    for x, (k1, b1, k2, b2) in zip((xz, yz, xy), conv_weights):
        x = np.asarray(x, np.float64)
        if x.ndim == 3:
            x = x[..., None]
        outs.append(keras_conv2d_same_s2_relu(keras_conv2d_same_s2_relu(x, k1, b1), k2, b2))
    cat = np.concatenate(outs, axis=-1)
    return cat.reshape(cat.shape[0], -1)


def dnn_forward(xz, yz, xy, conv_weights, dense_weights):
    """define_classifier forward at inference (dropout inactive): Dense 64 relu, Dense 64 relu, Dense n softmax
    (dnn.py:78-88).  dense_weights: [(kernel (in,out), bias)] * 3."""
    h = dnn_conv_features(xz, yz, xy, conv_weights)
    (w1, c1), (w2, c2), (w3, c3) = dense_weights
    h = np.maximum(h @ w1 + c1, 0.0)
    h = np.maximum(h @ w2 + c2, 0.0)
    z = h @ w3 + c3
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=1, keepdims=True)


# ------------------------------------------------------------------------------------------------------------------
# training-time augmentation: train.DataGenerator.flow -> augment, train.py:84-185, restated with the
# reference's own library calls (scipy.ndimage); the random draws are arguments
# ------------------------------------------------------------------------------------------------------------------
def aug_rotate(p, angle):
    """train.py:87-94: ndimage.rotate(p, angle, reshape=False), clamp to [0,1]."""
    from scipy import ndimage
    out = ndimage.rotate(np.asarray(p), angle, reshape=False)
    out[out > 1.0] = 1.0
    out[out < 0.0] = 0.0
    return out


def aug_clipped_zoom(img, zoom_factor):
    """train.py:96-144: zoom out = ndimage.zoom pasted into the centre of a zero plane; zoom in = ndimage.zoom of the centre
    crop trimmed to the input size; clamp to [0,1]."""
    from scipy import ndimage
    img = np.asarray(img)
    h, w = img.shape[:2]
    if zoom_factor < 1:
        zh, zw = int(np.round(h * zoom_factor)), int(np.round(w * zoom_factor))
        top, left = (h - zh) // 2, (w - zw) // 2
        out = np.zeros_like(img)
        out[top:top + zh, left:left + zw] = ndimage.zoom(img, (zoom_factor, zoom_factor))
    elif zoom_factor > 1:
        zh, zw = int(np.ceil(h / zoom_factor)), int(np.ceil(w / zoom_factor))
        top, left = (h - zh) // 2, (w - zw) // 2
        out = ndimage.zoom(img[top:top + zh, left:left + zw], (zoom_factor, zoom_factor))
        tt, tl = (out.shape[0] - h) // 2, (out.shape[1] - w) // 2
        out = out[tt:tt + h, tl:tl + w]
    else:
        out = img.copy()
    out[out > 1.0] = 1.0
    out[out < 0.0] = 0.0
    return out


def aug_sparse_noise(q, draw):
    """train.py:146-154: ONE Gaussian draw added to the non-zero entries (sparsity kept), clamp to [0,1]."""
    qc = np.asarray(q).copy()
    qc[qc != 0] += float(draw)      # a Python float, as Generator.normal() returns: the float32 array stays float32 (added in float32)
    qc[qc > 1.0] = 1.0
    qc[qc < 0.0] = 0.0
    return qc


def aug_repetitions(labels, balance=True):
    """train.py:187-196 + 158: class weight = count of the most common class / count (1 without balancing);
    a sample is augmented int(np.round(weight)) times."""
    import collections
    mc = collections.Counter(labels).most_common()
    cw = {c: (mc[0][1] / cnt if balance else 1) for c, cnt in mc}
    return [int(np.round(cw[y])) for y in labels]
