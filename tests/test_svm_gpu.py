"""HIP SVM kernels against scikit-learn's golden outputs and the float64 oracle.

Bar (BASELINE.json north_star): class labels bit-exact, decision-function scores within 1e-5."""
import numpy as np
import pytest
import torch

import oracle_np as O
from conftest import load_golden, svm_model_arrays

pytestmark = pytest.mark.gpu

TOL = 1e-5       # north_star: "decision-function scores within 1e-5"
TOL_F32 = 5e-3   # explicit RML_PATH_F32 (f32 MFMA accumulate) is an opt-in approximate path


def _tol(m, ref, path="auto"):
    """1e-5 for the RBF decision values (K <= 1).  A linear-kernel SVC has |K| ~ 1e2: sklearn itself sees
    float32-rounded inputs (relative 6e-8 per value), so the comparison is relative there."""
    t = TOL_F32 if path == "f32" else TOL
    if m["kernel"] == "linear":
        t = max(t, 5e-6 * float(np.abs(ref).max()))
    return t


def _model(rml, g, **kw):
    m = svm_model_arrays(g)
    return rml.GpuSVC(m["sv"], m["dual_coef"], m["intercept"], m["n_support"], m["gamma"], m["classes"],
                      kernel=m["kernel"], calib_a=m["calib_a"], calib_b=m["calib_b"], **kw), m


def _test_rows(g, name):
    if name == "real_xy_svm.npz":
        q = g["xy_u8"].reshape(len(g["xy_u8"]), -1)[g["test_idx"]]
    else:
        q = g["test_feat_u8"]
    return q.astype(np.float32) / np.float32(255.0)


@pytest.mark.parametrize("name", ["svm_small.npz", "svm_small_linear.npz", "svm_small_xy.npz", "svm_walabot.npz",
                                  "real_xy_svm.npz"])
@pytest.mark.parametrize("path", ["auto", "i8", "f64", "f32"])
def test_golden_parity_with_sklearn(rml, name, path):
    g = load_golden(name)
    svc, m = _model(rml, g, path=path)
    assert svc.exact and svc.code_scale == 255.0
    X = _test_rows(g, name)
    svc.decision_function_shape = "ovo"
    ovo = svc.decision_function(X)
    svc.decision_function_shape = "ovr"
    ovr = svc.decision_function(X)
    assert ovo.dtype == np.float64 and ovr.shape == g["dec_ovr"].shape
    tol = _tol(m, g["dec_ovo"], path)
    assert np.abs(ovo - g["dec_ovo"]).max() <= tol
    assert np.abs(ovr - g["dec_ovr"]).max() <= tol
    cal = rml.GpuCalibratedClassifier(svc)
    proba = cal.predict_proba(X)
    assert np.abs(proba - g["proba"]).max() <= tol
    if path != "f32":                                                         # bit-exact labels
        np.testing.assert_array_equal(svc.predict(X), g["label_vote"])
        np.testing.assert_array_equal(cal.predict(X), g["label_calib"])
    if path == "f64" and m["kernel"] == "rbf":
        # float64 products and sums of the very float32 inputs sklearn sees: float64 round-off only
        assert np.abs(ovo - g["dec_ovo"]).max() <= 1e-9
    if path == "i8" and m["kernel"] == "rbf":
        # exact integer distances: only float64 round-off and sklearn's own float32 inputs remain
        assert np.abs(ovo - g["dec_ovo"]).max() <= 2e-6


@pytest.mark.parametrize("path", ["auto", "i8", "f64"])
def test_binary_model(rml, path):
    """2-class problem (the reference's optional person/pet aliasing): sklearn flips the sign of the single
    libsvm pair value and uses one calibrator."""
    g = load_golden("svm_small_binary.npz")
    svc, m = _model(rml, g, path=path)
    X = _test_rows(g, "svm_small_binary.npz")
    dec = svc.decision_function(X)
    assert dec.shape == g["dec_ovr"].shape == (len(X),)
    assert np.abs(dec - g["dec_ovr"]).max() <= TOL
    np.testing.assert_array_equal(svc.predict(X), g["label_vote"])
    cal = rml.GpuCalibratedClassifier(svc)
    assert np.abs(cal.predict_proba(X) - g["proba"]).max() <= TOL
    np.testing.assert_array_equal(cal.predict(X), g["label_calib"])


def test_general_float_rows_vs_oracle(rml):
    """Non-integer features (e.g. augmented / zoomed data): auto -> f64 MFMA path vs the float64 oracle."""
    g = load_golden("svm_small.npz")
    svc, m = _model(rml, g)
    rng = np.random.default_rng(5)
    X = _test_rows(g, "svm_small.npz")
    X = np.clip(X + rng.normal(0, 0.02, X.shape).astype(np.float32) * (X > 0), 0, 1).astype(np.float32)
    want = O.svm_decision_ovo(X, m["sv"], m["dual_coef"], m["intercept"], m["n_support"], m["gamma"])
    svc.decision_function_shape = "ovo"
    got = svc.decision_function(X)
    assert np.abs(got - want).max() <= 1e-9
    C = len(m["classes"])
    np.testing.assert_array_equal(svc.predict(X), m["classes"][O.svm_vote_labels(want, C)])
    proba = rml.GpuCalibratedClassifier(svc).predict_proba(X)
    wantp = O.calibrated_proba(O.ovr_decision_function(want, C), m["calib_a"], m["calib_b"])
    assert np.abs(proba - wantp).max() <= TOL


def _irregular_rows(rng, X, kind):
    """Rows off the code grid the way the reference produces them: spline-resampled (train.py:96-144 zoom / rotate give
    arbitrary float32 values, a few slightly outside [0, 1] before the clamp), sparse noise (train.py:146-160), and
    values that are tiny next to the range (the spline tails)."""
    X = X.astype(np.float32).copy()
    if kind == "smooth":
        X = X * np.float32(0.9990234375) + rng.normal(0, 1e-3, X.shape).astype(np.float32) * (X > 0)
    elif kind == "noise":
        mask = rng.random(X.shape) < 0.05
        X = np.clip(X + mask * rng.normal(0, 0.05, X.shape).astype(np.float32), 0, 1).astype(np.float32)
    elif kind == "tails":
        X = X + (rng.random(X.shape) < 0.3) * rng.uniform(-3e-5, 3e-5, X.shape).astype(np.float32)
    return X.astype(np.float32)


@pytest.mark.parametrize("name", ["svm_small.npz", "svm_walabot.npz"])
@pytest.mark.parametrize("kind", ["smooth", "noise", "tails"])
@pytest.mark.parametrize("model_on_grid", [True, False])
def test_multi_digit_path_vs_oracle(rml, name, kind, model_on_grid):
    """RML_PATH_DIGITS: general rows as four int8 digits of a 32-bit fixed-point value, ten exact digit-plane GEMMs
    (csrc/svm.hip k_svm_gemm_dig) against the float64 oracle on the same float32 rows: decision values within 1e-5
    (measured ~1e-8), labels bit-exact, at D = 368 and D = 10 010 (the Walabot grid)."""
    g = load_golden(name)
    m = svm_model_arrays(g)
    rng = np.random.default_rng(11)
    sv = m["sv"]
    if not model_on_grid:
        sv = _irregular_rows(rng, sv, "smooth").astype(np.float64)
    svc = rml.GpuSVC(sv, m["dual_coef"], m["intercept"], m["n_support"], m["gamma"], m["classes"],
                     calib_a=m["calib_a"], calib_b=m["calib_b"], path="digits")
    assert svc.exact == model_on_grid
    X = _irregular_rows(rng, np.tile(_test_rows(g, name), (3, 1))[:300], kind)
    want = O.svm_decision_ovo(X, sv, m["dual_coef"], m["intercept"], m["n_support"], m["gamma"])
    svc.decision_function_shape = "ovo"
    got = svc.decision_function(X)
    err = float(np.abs(got - want).max())
    print("digits %s %s grid=%s: max |dec - oracle| = %.2e" % (name, kind, model_on_grid, err))
    assert err <= 1e-6, err
    C = len(m["classes"])
    np.testing.assert_array_equal(svc.predict(X), m["classes"][O.svm_vote_labels(want, C)])
    proba = rml.GpuCalibratedClassifier(svc).predict_proba(X)
    wantp = O.calibrated_proba(O.ovr_decision_function(want, C), m["calib_a"], m["calib_b"])
    assert np.abs(proba - wantp).max() <= TOL
    # and the float64 MFMA path agrees with it to the digit path's own resolution
    f64 = svc._decide(svc._rows(X), path="f64")[0].cpu().numpy()
    assert np.abs(got - f64).max() <= 1e-6


def test_multi_digit_rows_outside_the_fixed_point_range_fall_back(rml):
    """A tile with a row outside the model's fixed-point frame (|v - c0| >= s) is decided by the float64 kernel; the other
    tiles of the batch stay on the digit kernel; every row is right either way."""
    g = load_golden("svm_small.npz")
    m = svm_model_arrays(g)
    svc = rml.GpuSVC(m["sv"], m["dual_coef"], m["intercept"], m["n_support"], m["gamma"], m["classes"], path="digits")
    rng = np.random.default_rng(3)
    X = _irregular_rows(rng, np.tile(_test_rows(g, "svm_small.npz"), (9, 1))[:700], "smooth")
    X[5, 7] = 3.5            # first 256-row group: out of range -> float64 kernel
    X[600, 0] = np.float32(-0.4)   # still inside [-1, 1) * s around c0
    want = O.svm_decision_ovo(X, m["sv"], m["dual_coef"], m["intercept"], m["n_support"], m["gamma"])
    svc.decision_function_shape = "ovo"
    got = svc.decision_function(X)
    assert np.abs(got - want).max() <= 1e-6
    assert np.abs(got[:256] - want[:256]).max() <= 1e-9        # that group went through float64 products


def test_multi_digit_adversarial_low_digits(rml):
    """Worst case for the dropped digit pairs (i + j >= 4): every low digit at its extreme and aligned in sign.  The
    documented bound on |d(u.u)| is 3 * 2^-46 * 2^14 * D; the decision values must still be inside 1e-5 here."""
    D, M, N = 4096, 256, 256
    rng = np.random.default_rng(7)
    # values whose fixed-point digits below the top one are all +127 / -128 (c0 = 0.5, s = 1 for SVs spanning [0, 1])
    base = rng.integers(1, 120, (M + N, D)).astype(np.float64) / 256.0          # top digit
    low = (127 * 2.0 ** -15 + 127 * 2.0 ** -23 + 127 * 2.0 ** -31)
    vals = base + low
    sv = vals[:M].copy(); sv[0, 0] = 0.0; sv[0, 1] = 1.0                           # pin the range to [0, 1]
    X = vals[M:].astype(np.float32)
    ns = np.array([M // 2, M - M // 2], dtype=np.int32)
    dc = rng.uniform(-10, 10, (1, M))
    svc = rml.GpuSVC(sv, dc, np.array([0.1]), ns, 0.01, np.arange(2), path="digits")
    want = O.svm_decision_ovo(X, sv, dc, np.array([0.1]), ns, 0.01)
    svc.decision_function_shape = "ovo"
    got = svc._decide(svc._rows(X))[0].cpu().numpy()
    assert np.abs(got - want).max() <= TOL, np.abs(got - want).max()


def test_non_grid_model_uses_f64_path(rml):
    g = load_golden("svm_small.npz")
    m = svm_model_arrays(g)
    sv = m["sv"] * 1.0000001            # no longer on the k/255 grid
    svc = rml.GpuSVC(sv, m["dual_coef"], m["intercept"], m["n_support"], m["gamma"], m["classes"])
    assert not svc.exact
    X = _test_rows(g, "svm_small.npz")
    want = O.svm_decision_ovo(X, sv, m["dual_coef"], m["intercept"], m["n_support"], m["gamma"])
    svc.decision_function_shape = "ovo"
    # these SVs are not float32-representable; the device copy is float32 (relative 6e-8 per element)
    assert np.abs(svc.decision_function(X) - want).max() <= TOL
    with pytest.raises(rml.RadarMLError):
        svc._decide(svc._rows(X), path="i8")


@pytest.mark.parametrize("n", [1, 127, 128, 129, 300])
def test_ragged_batch_sizes(rml, n):
    g = load_golden("svm_small.npz")
    svc, m = _model(rml, g)
    X = np.tile(_test_rows(g, "svm_small.npz"), (2, 1))[:n]
    ref = np.tile(g["dec_ovr"], (2, 1))[:n]
    assert np.abs(svc.decision_function(X) - ref).max() <= TOL
    np.testing.assert_array_equal(svc.predict(X), np.tile(g["label_vote"], 2)[:n])


def test_input_validation(rml):
    g = load_golden("svm_small.npz")
    svc, m = _model(rml, g)
    with pytest.raises(ValueError):
        svc.predict(np.zeros((3, 5), np.float32))        # wrong D (sklearn raises ValueError too)
    with pytest.raises(ValueError):
        svc.predict(np.zeros(368, np.float32))
    out = svc.decision_function(np.zeros((0, 368), np.float32))
    assert out.shape == (0, 3)
    X64 = _test_rows(g, "svm_small.npz").astype(np.float64)        # any real dtype is accepted
    np.testing.assert_array_equal(svc.predict(X64), g["label_vote"])


@pytest.mark.parametrize("name,shape", [("svm_small.npz", (8, 10, 16)), ("svm_walabot.npz", (22, 31, 176)),
                                        ("svm_small_xy.npz", (8, 10, 16))])
def test_fused_volumes_to_labels(rml, name, shape):
    """volumes -> max-projection -> SVM in one call == sklearn on the same frames."""
    g = load_golden(name)
    svc, m = _model(rml, g)
    vol = g["test_vol_u8"].astype(np.float32)
    mask = rml.ProjMask(*[bool(b) for b in g["mask"]])
    out = svc.decide_volumes(vol, mode="max", proj_mask=mask, scale=True)
    assert np.abs(out["dec_ovo"].cpu().numpy() - g["dec_ovo"]).max() <= TOL
    assert np.abs(out["dec_ovr"].cpu().numpy() - g["dec_ovr"]).max() <= TOL
    assert np.abs(out["proba"].cpu().numpy() - g["proba"]).max() <= TOL
    np.testing.assert_array_equal(m["classes"][out["label_vote"].cpu().numpy()], g["label_vote"])
    np.testing.assert_array_equal(m["classes"][out["label_calib"].cpu().numpy()], g["label_calib"])
    # the fixture's volumes as they are stored (uint8): the same bits as the float32 ingest
    out8 = svc.decide_volumes(g["test_vol_u8"], mode="max", proj_mask=mask, scale=True)
    for k in ("dec_ovo", "dec_ovr", "proba", "label_vote", "label_calib"):
        assert torch.equal(out8[k], out[k]), k
    # and the two halves separately: code rows from the projection, then the GEMM + finish alone
    _, q, isum, isq, flags = rml.process_volumes(torch.from_numpy(vol).cuda(), mode="max", proj_mask=mask, scale=True, codes=True)
    ovo_c, ovr_c, vote_c, proba_c, lab_c = svc.decide_codes(q, isum, isq, flags, want_proba=True)
    assert torch.equal(ovo_c, out["dec_ovo"]) and torch.equal(vote_c, out["label_vote"]) and torch.equal(lab_c, out["label_calib"])
    # a frame with a non-integer return drops its tile to the f32 path; results stay within tolerance
    vol2 = vol.copy()
    vol2[3, 0, 0, 0] = 0.5
    out2 = svc.decide_volumes(vol2, mode="max", proj_mask=mask, scale=True)
    d = np.abs(out2["dec_ovo"].cpu().numpy() - g["dec_ovo"])
    d[3] = 0
    assert d.max() <= TOL


def test_fused_multi_chunk_deterministic(rml):
    """> 1 chunk (4096 frames each) through the two-stream pipeline, twice: identical bits."""
    import torch
    g = load_golden("svm_small.npz")
    svc, m = _model(rml, g)
    B = 4096 * 2 + 300
    v, _ = rml.synth_volumes(B, 8, 10, 16, seed=11)
    a = svc.decide_volumes(v)
    b = svc.decide_volumes(v)
    for k in a:
        assert torch.equal(a[k], b[k])
    # against separate projection + decision on a slab
    feat = rml.process_volumes(v[4000:4400], scale=True)
    svc.decision_function_shape = "ovo"
    sep = svc.decision_function(feat)
    np.testing.assert_array_equal(a["dec_ovo"][4000:4400].cpu().numpy(), sep)
    want = O.svm_decision_ovo(feat.cpu().numpy(), m["sv"], m["dual_coef"], m["intercept"], m["n_support"], m["gamma"])
    assert np.abs(sep - want).max() <= TOL
    np.testing.assert_array_equal(a["label_vote"][4000:4400].cpu().numpy(), O.svm_vote_labels(want, 3))


@pytest.mark.parametrize("name,shape", [("svm_small.npz", (8, 10, 16)), ("svm_walabot.npz", (22, 31, 176))])
@pytest.mark.parametrize("u8", [False, True])
@pytest.mark.parametrize("derive", [False, True])
def test_read_compare_write_of_the_code_rows_changes_nothing(rml, name, shape, u8, derive, rml_opt):
    """ProjOut::q_rmw (rml_internal.h): the chunk workspaces' code rows are read and only the words that changed are stored.
    Whatever the workspace held -- here the rows of a different batch, of the same batch, and of a batch with frames off the
    code grid -- the outputs are the bits of the plain stores."""
    g = load_golden(name)
    svc, m = _model(rml, g)
    mask = rml.ProjMask(*[bool(b) for b in g["mask"]])
    X, Y, Z = shape
    B = 4096 * 2 + 77                                   # three chunks: two workspaces in rotation
    batches = []
    for seed in (3, 4):
        v, _ = rml.synth_volumes(B, X, Y, Z, seed=seed)
        batches.append(v.to(torch.uint8) if u8 else v)
    if not u8:
        off = batches[1].clone()
        off[5, 1, 2, 3] = 0.25                          # its tile leaves the exact path
        batches.append(off)
    kw = dict(mode="slice") if derive else dict(mode="max")      # slice without ijk: the fused derive -> slice -> SVM pass
    want = []
    rml_opt("code_rmw", 0)
    for v in batches:
        want.append({k: t.clone() for k, t in svc.decide_volumes(v, proj_mask=mask, scale=True, **kw).items()})
    rml_opt("code_rmw", 1)
    for order in ((0, 1, 1, 0), (2, 0, 2) if not u8 else (1, 0)):
        for i in order:
            got = svc.decide_volumes(batches[i], proj_mask=mask, scale=True, **kw)
            for k in want[i]:
                assert torch.equal(got[k], want[i][k]), (k, i)


def test_linear_classifier_golden(rml):
    g = load_golden("linear_golden.npz")
    clf = rml.GpuLinearClassifier(g["coef"], g["intercept"], g["classes"], g["calib_a"], g["calib_b"])
    X = g["test_feat_u8"].astype(np.float32) / np.float32(255.0)
    assert np.abs(clf.decision_function(X) - g["dec"]).max() <= 1e-9
    np.testing.assert_array_equal(clf.predict(X), g["label"])
    cal = rml.GpuCalibratedClassifier(clf)
    assert np.abs(cal.predict_proba(X) - g["proba"]).max() <= 1e-9
    np.testing.assert_array_equal(cal.predict(X), g["label_calib"])


def test_linear_fused_volumes(rml):
    g = load_golden("linear_golden.npz")
    clf = rml.GpuLinearClassifier(g["coef"], g["intercept"], g["classes"], g["calib_a"], g["calib_b"])
    vol, _ = O.synth_volumes(1234 + 77, 700, 10, 12, 16)           # the frames the golden model was fitted on
    out = clf.decide_volumes(vol[600:700], mode="max", scale=True)
    assert np.abs(out["dec"].cpu().numpy() - g["dec"]).max() <= 1e-9
    np.testing.assert_array_equal(g["classes"][out["label"].cpu().numpy()], g["label"])
    np.testing.assert_array_equal(g["classes"][out["label_calib"].cpu().numpy()], g["label_calib"])


def test_from_sklearn_roundtrip(rml):
    sklearn = pytest.importorskip("sklearn")
    import warnings
    from sklearn import svm
    from sklearn.calibration import CalibratedClassifierCV
    vol, cls = O.synth_volumes(21, 500, 8, 10, 16)
    xz, yz, xy = O.project_max(vol)
    F = O.features_from_projections(xz, yz, xy, scale=True)
    clf = svm.SVC(kernel="rbf", C=10, gamma=0.05, class_weight="balanced").fit(F[:350], cls[:350])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cal = CalibratedClassifierCV(estimator=clf, cv="prefit").fit(F[350:420], cls[350:420])
    gpu = rml.from_sklearn(cal)
    Xt = F[420:]
    np.testing.assert_array_equal(gpu.predict(Xt), cal.predict(Xt))
    assert np.abs(gpu.predict_proba(Xt) - cal.predict_proba(Xt)).max() <= TOL
    assert np.abs(gpu.decision_function(Xt) - clf.decision_function(Xt)).max() <= TOL
    np.testing.assert_array_equal(rml.from_sklearn(clf).predict(Xt), clf.predict(Xt))
    # predict.classifier surface (predict.py:56-70)
    class LE:
        classes_ = np.array(["cat", "dog", "person"])
    name, p = rml.classifier(Xt[0], gpu, LE())
    pr = cal.predict_proba(Xt[:1])[0]
    assert abs(p - pr.max()) <= TOL and name == (LE.classes_[pr.argmax()] if pr.max() >= 0.7 else "Unknown")


def test_full_size_properties_64x64x128(rml):
    """BASELINE configs[2] shape (64x64x128, D = 20 480, M ~ 2k SVs) at > 1 chunk: size-independent properties of
    the fused path -- determinism, permutation equivariance, independence of the chunk boundaries -- plus a slab
    against the float64 oracle."""
    import torch
    import oracle_c as OC
    B, X, Y, Z = 8192 + 1500, 64, 64, 128
    V, _ = rml.synth_volumes(B, X, Y, Z, seed=99)
    feat, q, isum, isq, flags = rml.process_volumes(V[:2048], mode="max", scale=True, codes=True)
    M = 2048
    svq = (q[:M, :20480] ^ 0x80).cpu().numpy()
    sv = (svq.astype(np.float32) / np.float32(255.0)).astype(np.float64)
    rng = np.random.default_rng(3)
    ns = np.array([700, 700, 648], dtype=np.int32)
    dc = rng.uniform(-10, 10, (2, M)); ic = np.array([0.3, -0.1, 0.2])
    ca = np.array([-2.0, -1.5, -2.5]); cb = np.array([0.1, 0.0, -0.2])
    svc = rml.GpuSVC(sv, dc, ic, ns, 0.01, np.arange(3), calib_a=ca, calib_b=cb)
    assert svc.exact
    a = svc.decide_volumes(V)
    b = svc.decide_volumes(V)
    for k in a:
        assert torch.equal(a[k], b[k])                                   # deterministic
    perm = torch.randperm(B, device=V.device, generator=torch.Generator(device=V.device).manual_seed(0))
    c = svc.decide_volumes(V[perm].contiguous())
    for k in a:
        assert torch.equal(c[k], a[k][perm])                             # frames are independent
    d1 = svc.decide_volumes(V[:5000].contiguous()); d2 = svc.decide_volumes(V[5000:].contiguous())
    for k in a:
        assert torch.equal(torch.cat([d1[k], d2[k]]), a[k])              # chunk boundaries do not matter
    # supports vectors classify themselves consistently: frames 0..M-1 ARE the SVs -> K(x_m, sv_m) = 1 exactly
    n = 96
    vh = V[3000:3000 + n].cpu().numpy()
    xz, yz, xy = OC.project_max(vh, threads=8)
    ref = OC.svm(OC.features(xz, yz, xy, scale=True), sv, dc, ic, ns, 0.01, "rbf", ca, cb, threads=8)
    assert np.abs(a["dec_ovo"][3000:3000 + n].cpu().numpy() - ref["dec_ovo"]).max() <= 1e-5
    assert np.abs(a["proba"][3000:3000 + n].cpu().numpy() - ref["proba"]).max() <= 1e-5
    np.testing.assert_array_equal(a["label_vote"][3000:3000 + n].cpu().numpy(), ref["label_vote"])
    np.testing.assert_array_equal(a["label_calib"][3000:3000 + n].cpu().numpy(), ref["label_calib"])


@pytest.mark.parametrize("tag,C", [("c3", 3), ("c2", 2)])
def test_svc_predict_proba_platt(rml, tag, C):
    """SVC(probability=True).predict_proba: libsvm Platt scaling + pairwise coupling on the GPU vs sklearn."""
    g = load_golden("svm_platt.npz")
    sv = (g[tag + "_sv_u8"].astype(np.float32) / np.float32(255.0)).astype(np.float64)
    svc = rml.GpuSVC(sv, g[tag + "_dual_coef"], g[tag + "_intercept"], g[tag + "_n_support"], float(g["gamma"]),
                     g[tag + "_classes"], probA=g[tag + "_probA"], probB=g[tag + "_probB"], decision_function_shape="ovo")
    X = g["test_feat_u8"].astype(np.float32) / np.float32(255.0)
    assert np.abs(svc.decision_function(X) - g[tag + "_dec"]).max() <= TOL
    assert np.abs(svc.predict_proba(X) - g[tag + "_proba"]).max() <= TOL
    np.testing.assert_array_equal(svc.predict(X), g[tag + "_label_vote"])
    bare = rml.GpuSVC(sv, g[tag + "_dual_coef"], g[tag + "_intercept"], g[tag + "_n_support"], float(g["gamma"]), g[tag + "_classes"])
    with pytest.raises(AttributeError):
        bare.predict_proba(X)


def test_slice_mode_fused_with_derived_targets(rml):
    """Reference-faithful pipeline without the Walabot SDK: derive (i,j,k) on the GPU (common.py:49-80), slice the
    planes there (predict.py:102-107), scale, classify -- all in one call, against the CPU oracle."""
    import oracle_c as OC
    g = load_golden("svm_walabot.npz")
    svc, m = _model(rml, g)
    vol = g["test_vol_u8"].astype(np.float32)
    out = svc.decide_volumes(vol, mode="slice", scale=True)
    ijk = np.array([[t.i, t.j, t.k] for v in vol for t in O.get_derived_targets(v, 22, 31, 176)])
    xz, yz, xy = OC.project_slice(vol, ijk)
    ref = OC.svm(OC.features(xz, yz, xy, scale=True), m["sv"], m["dual_coef"], m["intercept"], m["n_support"], m["gamma"],
                 "rbf", m["calib_a"], m["calib_b"], threads=2)
    assert np.abs(out["dec_ovo"].cpu().numpy() - ref["dec_ovo"]).max() <= TOL
    np.testing.assert_array_equal(out["label_calib"].cpu().numpy(), ref["label_calib"])
    feat = rml.process_volumes(vol, mode="slice", scale=True).cpu().numpy()
    np.testing.assert_array_equal(feat, OC.features(xz, yz, xy, scale=True))


def test_kernel_matrix_service(rml):
    """rml_svm_kernel_matrix: K(X, SV) equals the oracle's float64 kernel values (exact-integer path: <= 1e-12
    relative; non-integer rows through the float64 MFMA path), and an SVC fitted on the GPU Gram matrix with
    kernel='precomputed' predicts exactly like the SVC fitted by libsvm itself (train.py:462-491 workload)."""
    from sklearn.svm import SVC
    g = load_golden("svm_small.npz")
    svc, m = _model(rml, g)
    X = _test_rows(g, "svm_small.npz")
    K = svc.kernel_matrix(X).cpu().numpy()
    want = O.svm_kernel_values(X.astype(np.float64), m["sv"], m["gamma"])
    assert K.shape == want.shape == (len(X), svc.n_sv)
    # the exact path works on the integer codes; the oracle (like sklearn) on their float32-rounded c/255: <= 1e-7 * gamma * d^2
    assert np.abs(K - want).max() <= 5e-7
    Xn = X.copy()
    Xn[::3] += np.float32(0.001)                                    # rows off the code grid -> f64 path for their tiles
    Kn = svc.kernel_matrix(Xn).cpu().numpy()
    assert np.abs(Kn - O.svm_kernel_values(Xn.astype(np.float64), m["sv"], m["gamma"])).max() <= 5e-7
    assert np.abs(svc.kernel_matrix(X, path="i8").cpu().numpy() - K).max() == 0.0
    assert np.abs(svc.kernel_matrix(X, path="f64").cpu().numpy() - want).max() <= 1e-9      # same float32 rows as the oracle
    # Gram-matrix service: fit on K(train, train), predict with K(test, train)
    rng = np.random.default_rng(5)
    Xtr = (rng.integers(0, 256, (300, X.shape[1])).astype(np.float32) / np.float32(255.0))
    ytr = rng.integers(0, 3, 300)
    Xtr[ytr == 1, :40] = 1.0
    Xtr[ytr == 2, 40:80] = 0.0
    km = rml.KernelMatrix(Xtr, gamma=0.05)
    assert km.exact
    G = km.gram().cpu().numpy()
    assert np.allclose(np.diag(G), 1.0) and np.abs(G - G.T).max() == 0.0
    direct = SVC(kernel="rbf", gamma=0.05, C=10.0).fit(Xtr.astype(np.float64), ytr)
    pre = SVC(kernel="precomputed", C=10.0).fit(G, ytr)
    Kt = km.against(X[:64]).cpu().numpy()
    np.testing.assert_array_equal(pre.predict(Kt), direct.predict(X[:64].astype(np.float64)))
    assert np.abs(pre.decision_function(Kt) - direct.decision_function(X[:64].astype(np.float64))).max() < 1e-3


def test_more_than_four_classes_match_sklearn(rml):
    """5-class SVC (ovo / ovr / vote / calibrated proba / libsvm Platt coupling) and 6-class SGD against scikit-learn
    golden vectors (tests/golden/make_golden_multiclass.py); the reference's label set is open (train.py:656-663)."""
    g = load_golden("svm_multiclass.npz")
    sv = (g["svc_sv_u8"].astype(np.float32) / np.float32(255.0)).astype(np.float64)
    X = g["svc_test_u8"].astype(np.float32) / np.float32(255.0)
    for path in ("auto", "f64"):
        svc = rml.GpuSVC(sv, g["svc_dual_coef"], g["svc_intercept"], g["svc_n_support"], float(g["svc_gamma"]), g["svc_classes"],
                         calib_a=g["svc_calib_a"], calib_b=g["svc_calib_b"], probA=g["svc_probA"], probB=g["svc_probB"], path=path)
        svc.decision_function_shape = "ovo"
        assert np.abs(svc.decision_function(X) - g["svc_dec_ovo"]).max() <= TOL
        svc.decision_function_shape = "ovr"
        assert np.abs(svc.decision_function(X) - g["svc_dec_ovr"]).max() <= TOL
        np.testing.assert_array_equal(svc.predict(X), g["svc_label_vote"])
        cal = rml.GpuCalibratedClassifier(svc)
        assert np.abs(cal.predict_proba(X) - g["svc_proba"]).max() <= TOL
        np.testing.assert_array_equal(cal.predict(X), g["svc_label_calib"])
        assert np.abs(svc.predict_proba(X) - g["svc_platt_proba"]).max() <= TOL
    Xs = g["sgd_test_u8"].astype(np.float32) / np.float32(255.0)
    lin = rml.GpuLinearClassifier(g["sgd_coef"], g["sgd_intercept"], g["sgd_classes"], g["sgd_calib_a"], g["sgd_calib_b"])
    assert np.abs(lin.decision_function(Xs) - g["sgd_dec"]).max() <= 1e-9
    np.testing.assert_array_equal(lin.predict(Xs), g["sgd_label"])
    cal = rml.GpuCalibratedClassifier(lin)
    assert np.abs(cal.predict_proba(Xs) - g["sgd_proba"]).max() <= 1e-9
    np.testing.assert_array_equal(cal.predict(Xs), g["sgd_label_calib"])
    with pytest.raises(NotImplementedError):
        rml.GpuLinearClassifier(np.zeros((7, 4)), np.zeros(7), np.arange(7))


def test_from_sklearn_reads_objects_pickled_by_sklearn_024(rml):
    """A model written by the reference's own environment (scikit-learn 0.24, requirements.txt:57) carries
    base_estimator / calibrators_ on the calibrated classifier and probA_ / probB_ on the SVC: from_sklearn must load it
    through the documented drop-in edit.  The 0.24 layout is emulated by renaming the attributes of
    objects fitted here."""
    import types
    import warnings
    from sklearn import svm
    from sklearn.calibration import CalibratedClassifierCV
    g = load_golden("svm_small.npz")
    m = svm_model_arrays(g)
    rng = np.random.default_rng(5)
    Xtr = rng.integers(0, 256, (240, 40)).astype(np.float32) / np.float32(255.0)
    ytr = rng.integers(0, 3, 240)
    clf = svm.SVC(kernel="rbf", C=10.0, gamma=0.05, probability=True, random_state=1).fit(Xtr, ytr)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cal = CalibratedClassifierCV(estimator=clf, cv="prefit").fit(Xtr[:90], ytr[:90])
    want_p, want_l = cal.predict_proba(Xtr), cal.predict(Xtr)
    # emulate the 0.24 pickle: private Platt arrays under their old public names, old calibrated-classifier layout
    SVC = type("SVC", (), {})                                       # from_sklearn dispatches on the class NAME
    CalibratedClassifierCV_ = type("CalibratedClassifierCV", (), {})
    old_svc = SVC()
    old_svc.__dict__.update({k: v for k, v in clf.__dict__.items() if k not in ("_probA", "_probB")})
    old_svc.probA_, old_svc.probB_ = clf._probA, clf._probB
    cc = cal.calibrated_classifiers_[0]
    old_cc = types.SimpleNamespace(base_estimator=old_svc, calibrators_=cc.calibrators, classes=cc.classes, method="sigmoid")
    old_cal = CalibratedClassifierCV_()
    old_cal.calibrated_classifiers_, old_cal.classes_ = [old_cc], cal.classes_
    gpu = rml.from_sklearn(old_cal)
    assert np.abs(gpu.predict_proba(Xtr) - want_p).max() <= TOL
    np.testing.assert_array_equal(gpu.predict(Xtr), want_l)
    assert np.abs(gpu.estimator.predict_proba(Xtr) - clf.predict_proba(Xtr)).max() <= TOL      # libsvm Platt from probA_/probB_


def test_pairwise_proba_is_asynchronous_and_needs_platt_coefficients(rml):
    import ctypes as C
    from radar_ml_amd import _lib
    g = load_golden("svm_platt.npz")
    sv = (g["c3_sv_u8"].astype(np.float32) / np.float32(255.0)).astype(np.float64)
    bare = rml.GpuSVC(sv, g["c3_dual_coef"], g["c3_intercept"], g["c3_n_support"], float(g["gamma"]), g["c3_classes"])
    lib = _lib.load()
    dec = torch.zeros((4, 3), dtype=torch.float64, device="cuda")
    out = torch.empty((4, 3), dtype=torch.float64, device="cuda")
    rc = lib.rml_svm_pairwise_proba(bare._ctx, bare._h, _lib.ptr(dec), 4, _lib.ptr(out), _lib.stream_ptr())
    assert rc == -5 and b"Platt" in lib.rml_last_error()            # RML_ERR_STATE
    with pytest.raises(AttributeError):
        bare.predict_proba(np.zeros((2, sv.shape[1]), np.float32))


@pytest.mark.parametrize("name", ["svm_small.npz", "svm_walabot.npz", "svm_small_linear.npz", "svm_small_binary.npz"])
def test_large_tile_exact_gemm_matches_the_oracle(rml, name, rml_opt):
    """k_svm_gemm_ring (256 SVs x 256 samples per workgroup, 5-slot operand-stage ring) is what large batches run on; forced
    here (RML_GEMM_BIG=1) on ragged batches: a row count that is no multiple of 256, an SV count that is no multiple of
    256, and -- through the float rows -- sample tiles that are NOT on the code grid, which must fall to the float64
    kernel pair-wise (tile flags are decided per 256 samples then).  Against the oracle, and BIT-identical to the 128 x 128
    kernel (RML_GEMM_BIG=0): the same int32 dot products, one float64 partial per 128 SV rows summed in the same order."""
    g = load_golden(name)
    svc, m = _model(rml, g)
    X0 = _test_rows(g, name)
    rng = np.random.default_rng(7)
    X = X0[rng.integers(0, len(X0), 1100)]
    X[300:420] *= np.float32(0.9990234375)             # rows 300..419 leave the code grid: tiles 2,3 (128-row units) -> pair 1
    C = len(m["classes"])
    want = O.svm_decision_ovo(X, m["sv"], m["dual_coef"], m["intercept"], m["n_support"], m["gamma"], m["kernel"])
    outs = {}
    for big in ("0", "1"):
        rml_opt("gemm_big", int(big))
        svc.decision_function_shape = "ovo"
        got = svc.decision_function(X)
        got = got.reshape(len(X), -1)
        ref = want if C > 2 else -want
        assert np.abs(got - ref).max() <= _tol(m, ref), big
        outs[big] = (got, svc.predict(X), rml.GpuCalibratedClassifier(svc).predict_proba(X))
    np.testing.assert_array_equal(outs["0"][1], outs["1"][1])
    # exact tiles give the same integers on both kernels, and the 256 x 256 kernel writes one partial per 128 SV rows summed
    # like the 128 x 128 kernel's: the decision values do not depend on the tile size (batch size, chunking, ingest dtype)
    np.testing.assert_array_equal(outs["0"][0], outs["1"][0])
    np.testing.assert_array_equal(outs["0"][2], outs["1"][2])
    # and on code rows straight from volumes (the fused pipeline's operand), labels identical between the kernels
    if name == "svm_walabot.npz":
        import torch
        vol = torch.from_numpy(np.tile(g["test_vol_u8"], (12, 1, 1, 1))[:700].astype(np.float32)).cuda()
        res = {}
        for big in ("0", "1"):
            rml_opt("gemm_big", int(big))
            o = svc.decide_volumes(vol, mode="max", scale=True, want_proba=True)
            res[big] = {k: v.cpu().numpy() for k, v in o.items()}
        for k in ("label_vote", "label_calib"):
            np.testing.assert_array_equal(res["0"][k], res["1"][k])
        np.testing.assert_array_equal(res["0"]["dec_ovo"], res["1"]["dec_ovo"])


@pytest.mark.parametrize("grid,frames", [((64, 64, 128), 65536), ((22, 31, 176), 262144)])
def test_full_size_batches_size_independent_properties(rml, grid, frames):
    """BASELINE.json's full sizes (configs[2]: 65 536 frames of 64x64x128; the Walabot grid at the bench's 262 144) through
    properties that do not need an oracle of that size:
      * chunking independence -- the whole batch in one call against the same frames in three ragged calls (the library chunks
        each call for itself: different chunk boundaries, different last-chunk sizes, 8 192-frame and whole-round paths): decision
        values, probabilities and labels BIT-identical;
      * ingest independence -- the same frames as uint8 volumes (byte kernel, ring GEMM in whole rounds): bit-identical again;
      * monotonicity of the max-projection -- features of max(V1, V2) = max of the features, on a slab;
      * the float64 C oracle on 96 frames drawn from the whole range (labels bit-exact, decision values within 1e-5).
    The model: support vectors on the code grid drawn from the same generator as the frames (an exact model, ~1.5 k SVs)."""
    X, Y, Z = grid
    import gc
    gc.collect(); torch.cuda.empty_cache()                 # earlier tests' blocks sit in PyTorch's caching allocator
    free = torch.cuda.mem_get_info()[0]
    need = frames * X * Y * Z * 5 + (8 << 30)               # float32 volumes + their uint8 copy + workspaces
    if free < need:
        pytest.skip("needs %.0f GB of free HBM" % (need / 2 ** 30))
    rng = np.random.default_rng(17)
    M = 1536
    svv, _ = rml.synth_volumes(M, X, Y, Z, seed=101)
    sv = rml.process_volumes(svv, mode="max", scale=True).cpu().numpy().astype(np.float64)
    del svv
    nsv = np.array([M // 3, M // 3, M - 2 * (M // 3)], dtype=np.int32)
    dual = rng.uniform(-1.0, 1.0, size=(2, M))
    icpt = rng.uniform(-0.1, 0.1, 3)
    svc = rml.GpuSVC(sv, dual, icpt, nsv, 0.01, np.array([0, 1, 2]),
                     calib_a=-np.abs(rng.uniform(1.0, 3.0, 3)), calib_b=rng.uniform(-0.2, 0.2, 3))
    assert svc.exact
    V, _ = rml.synth_volumes(frames, X, Y, Z, seed=7)
    whole = svc.decide_volumes(V, mode="max", scale=True)
    torch.cuda.synchronize()
    cuts = [0, frames // 3 + 1000, frames // 3 + 1000 + 12345, frames]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = svc.decide_volumes(V[lo:hi], mode="max", scale=True)
        for k in ("dec_ovo", "dec_ovr", "proba", "label_vote", "label_calib"):
            assert torch.equal(part[k], whole[k][lo:hi]), (k, lo, hi)
    # the same frames as bytes
    V8 = V.to(torch.uint8)
    bytes_out = svc.decide_volumes(V8, mode="max", scale=True)
    for k in ("dec_ovo", "proba", "label_calib"):
        assert torch.equal(bytes_out[k], whole[k]), k
    del V8, bytes_out
    # monotone: the projections of an element-wise maximum are the element-wise maximum of the projections
    a, b = V[:2048], V[frames - 2048:]
    fa, fb = rml.process_volumes(a, mode="max", scale=True), rml.process_volumes(b, mode="max", scale=True)
    fm = rml.process_volumes(torch.maximum(a, b), mode="max", scale=True)
    assert torch.equal(fm, torch.maximum(fa, fb))
    # the oracle on frames from the whole range
    pick = np.unique(np.concatenate([np.arange(32), rng.integers(0, frames, 32), np.arange(frames - 32, frames)]))
    idx = torch.as_tensor(pick, device=V.device)
    rows = O.features_from_projections(*O.project_max(V[idx].cpu().numpy()), (True, True, True), True).astype(np.float32)
    want = O.svm_decision_ovo(rows, sv, dual, icpt, nsv, 0.01, "rbf")
    assert np.abs(whole["dec_ovo"][idx].cpu().numpy() - want).max() <= 1e-5
    np.testing.assert_array_equal(whole["label_vote"][idx].cpu().numpy(), O.svm_vote_labels(want, 3))
    del V, whole, a, b, fa, fb, fm
    gc.collect(); torch.cuda.empty_cache()


@pytest.mark.parametrize("u8", [False, True])
def test_fused_front_door_edge_batch_sizes(rml, u8):
    """decide_volumes on batches of 0, 1, 127, 129 and 8 193 frames (empty; below, across and just past a 128-row tile; one frame
    past a pipeline chunk), float32 and uint8 volumes: every frame's outputs equal the ones it gets in the 8 193-frame batch."""
    g = load_golden("svm_walabot.npz")
    svc, m = _model(rml, g)
    X, Y, Z = 22, 31, 176
    V, _ = rml.synth_volumes(8193, X, Y, Z, seed=23)
    if u8:
        V = V.to(torch.uint8)
    whole = svc.decide_volumes(V, mode="max", scale=True)
    assert whole["dec_ovo"].shape == (8193, 3)
    for n in (0, 1, 127, 129):
        part = svc.decide_volumes(V[8193 - n:], mode="max", scale=True)
        for k in ("dec_ovo", "dec_ovr", "label_vote"):
            assert part[k].shape[0] == n
            assert torch.equal(part[k], whole[k][8193 - n:]), (k, n)


@pytest.mark.parametrize("grid,M", [((64, 64, 128), 2562), ((22, 31, 176), 700), ((8, 10, 16), 90)])
def test_single_observations_take_the_small_path_with_the_same_bits(rml, grid, M):
    """predict.py:98-119 classifies ONE observation per call.  Batches of 1..8 frames take the single-observation path (the frame
    split over the chip + k_project_finalize, k_svm_dot_small + k_svm_epi_small, everything on the caller's stream), batches whose
    128 x 128 tiles do not fill the machine (9 .. ~1 500 rows) the split-K tile kernel + the same chain epilogue: every output
    equals, bit for bit, what the same frame gets inside a 2 100-frame batch (the classic tile kernels); float32 and uint8 volumes,
    a projection mask, a frame off the code grid (float64 path for that call), decision values against the float64 oracle."""
    X, Y, Z = grid
    rng = np.random.default_rng(5)
    NW = 2100                                                       # 17 sample tiles x the model's SV tiles: more tiles than CUs
    V, _ = rml.synth_volumes(NW + M, X, Y, Z, seed=41)
    _, q, *_ = rml.process_volumes(V[NW:], mode="max", scale=True, codes=True)
    D = rml.feature_len(X, Y, Z)
    sv = ((q[:, :D] ^ 0x80).cpu().numpy().astype(np.float32) / np.float32(255.0)).astype(np.float64)
    ns = np.array([M // 3, M // 3, M - 2 * (M // 3)], dtype=np.int32)
    dual, icpt = rng.uniform(-3, 3, (2, M)), np.array([0.1, -0.2, 0.3])
    gamma = 4.0 / D
    svc = rml.GpuSVC(sv, dual, icpt, ns, gamma, np.arange(3), calib_a=np.array([-1.0, -1.1, -0.9]), calib_b=np.array([0.0, 0.1, -0.1]))
    V = V[:NW].contiguous()
    keys = ("dec_ovo", "dec_ovr", "proba", "label_vote", "label_calib")
    for vol in (V, V.to(torch.uint8)):
        whole = svc.decide_volumes(vol, mode="max", scale=True)
        for n, at in ((1, 0), (1, NW - 1), (2, 17), (3, 100), (5, 201), (8, 64), (9, 500), (64, 1000), (100, 1100), (128, 1300), (300, 1500), (1200, 700)):
            part = svc.decide_volumes(vol[at:at + n], mode="max", scale=True)
            for k in keys:
                assert torch.equal(part[k], whole[k][at:at + n]), (k, n, at, str(vol.dtype))
        again = svc.decide_volumes(vol[5:6], mode="max", scale=True)
        assert torch.equal(again["dec_ovo"], whole["dec_ovo"][5:6])
    # the rows API (SVC.decision_function on feature rows: train.py's clf.predict(X_test)): 1, 7, 150 rows against the 2 100
    feat = rml.process_volumes(V, mode="max", scale=True)
    svc.decision_function_shape = "ovo"
    dall = svc.decision_function(feat)
    for n, at in ((1, 3), (7, 40), (150, 333)):
        np.testing.assert_array_equal(svc.decision_function(feat[at:at + n]), dall[at:at + n])
    np.testing.assert_array_equal(dall, whole["dec_ovo"].cpu().numpy())
    # against the float64 oracle (one frame)
    xz, yz, xy = O.project_max(V[7:8].cpu().numpy())
    want = O.svm_decision_ovo(O.features_from_projections(xz, yz, xy, scale=True), sv, dual, icpt, ns, gamma)
    got = svc.decide_volumes(V[7:8], mode="max", scale=True)["dec_ovo"].cpu().numpy()
    assert np.abs(got - want).max() <= TOL
    # a projection mask (XZ + XY rows): the split projection still computes every plane, the row takes the ones asked for
    if grid == (8, 10, 16):
        Dm = rml.feature_len(X, Y, Z, rml.ProjMask(True, False, True))
        svm = rml.GpuSVC(np.concatenate([sv[:, :X * Z], sv[:, X * Z + Y * Z:]], axis=1), dual, icpt, ns, gamma, np.arange(3))
        assert Dm == X * Z + X * Y
        mask = rml.ProjMask(True, False, True)
        wm = svm.decide_volumes(V, mode="max", scale=True, proj_mask=mask)
        pm = svm.decide_volumes(V[33:36], mode="max", scale=True, proj_mask=mask)
        assert torch.equal(pm["dec_ovo"], wm["dec_ovo"][33:36]) and torch.equal(pm["label_vote"], wm["label_vote"][33:36])
    # slices at given voxels (the SDK target of predict.py:98-107), one observation per call: the same bits as inside the batch
    ijk = torch.stack([torch.randint(0, d, (NW,), generator=torch.Generator().manual_seed(3)) for d in (X, Y, Z)], dim=1).to(torch.int32).cuda()
    ws = svc.decide_volumes(V, mode="slice", ijk=ijk, scale=True)
    for n, at in ((1, 4), (3, 150), (8, 292), (40, 900)):
        ps = svc.decide_volumes(V[at:at + n], mode="slice", ijk=ijk[at:at + n], scale=True)
        for k in keys:
            assert torch.equal(ps[k], ws[k][at:at + n]), (k, n, at, "slice")
    # a frame off the code grid: that call takes the float64 path; within tolerance of the code-grid result of its neighbours' model
    v2 = V[10:12].clone()
    v2[0, 0, 0, 0] = 0.5
    off = svc.decide_volumes(v2, mode="max", scale=True)
    on = svc.decide_volumes(V[10:12], mode="max", scale=True)
    assert np.abs(off["dec_ovo"][1].cpu().numpy() - on["dec_ovo"][1].cpu().numpy()).max() <= TOL
    assert torch.equal(off["label_vote"][1], on["label_vote"][1])


def test_nan_row_through_the_svm_raises_like_scikit_learn(rml):
    """predict.py:60 -> clf.predict -> sklearn's validate_data: a row that holds a NaN or an infinity raises ValueError("Input contains
    NaN ...").  The sklearn-protocol methods of the three GPU twins do the same (round 5 returned finite garbage for such a row);
    the array-in / tensor-out entry (_decide, the fused decide_volumes) stays asynchronous and documents what it returns instead:
    the row is off the code grid, runs on the float64 path, and the rows next to it are untouched."""
    g = load_golden("svm_walabot.npz")
    svc, m = _model(rml, g)
    X = (g["test_feat_u8"][:9].astype(np.float32) / np.float32(255.0))
    clean = svc.decision_function(X)
    Xn = X.copy()
    Xn[4, 123] = np.nan
    Xi = X.copy()
    Xi[2, 7] = np.inf
    import sklearn.svm
    with pytest.raises(ValueError):
        sklearn.svm.SVC().fit(X[:4], [0, 1, 0, 1]).predict(Xn)         # the reference library refuses the row
    cal = rml.GpuCalibratedClassifier(svc)
    lin = rml.GpuLinearClassifier(np.zeros((3, X.shape[1])), np.zeros(3), np.arange(3))
    for bad in (Xn, Xi):
        for fn in (svc.predict, svc.decision_function, cal.predict, cal.predict_proba, lin.predict, lin.decision_function):
            with pytest.raises(ValueError, match="Input contains NaN"):
                fn(bad)
    np.testing.assert_array_equal(svc.decision_function(X), clean)      # ... and clean rows still pass
    ovo, ovr, vote, proba, lab = svc._decide(svc._rows(Xn, check_finite=False), want_proba=True)
    keep = [r for r in range(9) if r != 4]
    # (the tile holding the NaN row runs on the float64 kernel as a whole: its other rows agree with the exact path to rounding)
    np.testing.assert_allclose(ovr.cpu().numpy()[keep], clean[keep], rtol=0, atol=1e-6)


@pytest.mark.parametrize("grid", [(64, 31, 176), (40, 24, 192)])
def test_pipeline_on_wide_linear_plane_grids_matches_the_oracle(rml, grid):
    """a codes-only launch of k_project_lin with a large code stage -- 64 x 31 x 176 float32 asks for 67.5 KB of wave
    images + 4 x 12.9 KB of stage = 117.8 KB of dynamic LDS, past the 112 KB the kernel's attribute allowed (the launch failed with
    'invalid argument': the runtime does enforce it).  The attribute is the CU's 160 KB now, with a direct-store fallback beyond
    it.  The fused pipeline on such grids (one more of the linear-plane family: 48-quad rows) against the float64 oracle."""
    import oracle_np as O
    X, Y, Z = grid
    D = X * Z + Y * Z + X * Y
    rng = np.random.default_rng(5)
    M = 192
    sv_codes = (rng.random((M, D)) < 0.15) * rng.integers(13, 256, (M, D))
    sv = (sv_codes.astype(np.float32) / np.float32(255.0)).astype(np.float64)
    dual = rng.uniform(-3, 3, (2, M))
    icpt = rng.uniform(-0.5, 0.5, 3)
    nsup = np.array([64, 64, 64], dtype=np.int32)
    svc = rml.GpuSVC(sv, dual, icpt, nsup, 0.01, np.arange(3), calib_a=np.array([-1.5, -1.2, -1.8]), calib_b=np.array([0.1, 0.0, -0.1]))
    assert svc.exact
    vol, _ = O.synth_volumes(17, 640, X, Y, Z)                       # >= 2 * 256 frames: the persistent wave-per-frame kernels take it
    out = svc.decide_volumes(torch.from_numpy(vol).cuda(), mode="max", scale=True, want_proba=True)
    import oracle_c as OC
    xz, yz, xy = OC.project_max(vol, threads=8)
    ref = OC.svm(OC.features(xz, yz, xy, scale=True), sv, dual, icpt, nsup, 0.01, "rbf", np.array([-1.5, -1.2, -1.8]), np.array([0.1, 0.0, -0.1]),
                 threads=8)
    assert np.abs(out["dec_ovo"].cpu().numpy() - ref["dec_ovo"]).max() <= TOL
    np.testing.assert_array_equal(out["label_vote"].cpu().numpy(), ref["label_vote"])
    np.testing.assert_array_equal(out["label_calib"].cpu().numpy(), ref["label_calib"])
    # and the two-step route (code rows, then the GEMM) gives the same bits
    _, q, isum, isq, flags = rml.process_volumes(torch.from_numpy(vol).cuda(), mode="max", scale=True, codes=True)
    ovo_c, _, vote_c, _, lab_c = svc.decide_codes(q, isum, isq, flags, want_proba=True)
    assert torch.equal(ovo_c, out["dec_ovo"]) and torch.equal(vote_c, out["label_vote"]) and torch.equal(lab_c, out["label_calib"])
