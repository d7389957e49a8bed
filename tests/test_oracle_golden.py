"""The oracle (oracle/oracle_np.py) against the committed golden vectors.

The golden vectors were produced by the reference itself (common.py / predict.py imported in the
build container) and by scikit-learn, the reference's SVM implementation -- see
tests/golden/make_golden.py.  This pins the oracle; the GPU tests then compare the HIP path with
the oracle and with the same golden vectors.
"""
import numpy as np
import pytest

import oracle_np as O
from conftest import load_golden, svm_model_arrays


def test_index_known_answers_from_reference_log():
    g = load_golden("index_kats.npz")
    X, Y, Z = g["sizes"]
    assert len(g["log_xyz"]) == 28
    for (x, y, z), want in zip(g["log_xyz"], g["log_ijk"]):
        assert O.calculate_matrix_indices(x, y, z, X, Y, Z) == tuple(want)
    for (x, y, z), want in zip(g["rand_xyz"], g["rand_ijk"]):
        assert O.calculate_matrix_indices(x, y, z, X, Y, Z) == tuple(want)
    sph = np.array([O.cartesian_to_spherical(*p) for p in g["rand_xyz"]])
    np.testing.assert_array_equal(sph, g["rand_sph"])
    car = np.array([O.spherical_to_cartesian(*p) for p in sph])
    np.testing.assert_array_equal(car, g["rand_car"])


def test_derived_targets_and_slices_match_reference():
    g = load_golden("common_golden.npz")
    vol = g["volumes_u8"].astype(np.float32)
    X, Y, Z = vol.shape[1:]
    for nt in (1, 3):
        for b, v in enumerate(vol):
            t = O.get_derived_targets(v, X, Y, Z, num_targets=nt)
            np.testing.assert_array_equal(np.array([[d.i, d.j, d.k] for d in t]), g["derived_ijk_%d" % nt][b])
            np.testing.assert_allclose(np.array([[d.xPosCm, d.yPosCm, d.zPosCm] for d in t]),
                                       g["derived_xyz_%d" % nt][b], rtol=0, atol=0)


@pytest.mark.parametrize("mi", [0, 1, 2, 3])
@pytest.mark.parametrize("scale", [False, True])
def test_process_samples_matches_reference(mi, scale):
    g = load_golden("common_golden.npz")
    vol = g["volumes_u8"].astype(np.float32)
    mask = tuple(bool(x) for x in g["masks"][mi])
    samples = [O.project_slice(v, *ijk) for v, ijk in zip(vol, g["slice_ijk"])]
    want = g["feat_m%d_s%d" % (mi, int(scale))]
    got = O.process_samples(samples, proj_mask=O.ProjMask(*mask), scale=scale)
    assert got.dtype == want.dtype and got.shape == want.shape
    np.testing.assert_array_equal(got, want)          # same SciPy call -> identical
    # the identity-zoom restatement used by the HIP path agrees to the spline round-trip error
    ident = O.features_from_projections(np.array([s[0] for s in samples]), np.array([s[1] for s in samples]),
                                        np.array([s[2] for s in samples]), mask, scale)
    assert np.abs(ident - want).max() <= (2e-13 if not scale else 1e-15)


def test_process_samples_nonunit_zoom_matches_reference():
    g = load_golden("common_golden.npz")
    zf = g["zoom_factors"]
    zoom = O.ProjZoom(xz=list(zf[0]), yz=list(zf[1]), xy=list(zf[2]))
    samples = list(zip(g["zoom_in_xz"], g["zoom_in_yz"], g["zoom_in_xy"]))
    got = O.process_samples(samples, proj_zoom=zoom, scale=True)
    np.testing.assert_array_equal(got, g["zoom_feat"])
    z2 = O.calc_proj_zoom(22, 31, 176, 20, 28, 160)
    assert list(z2.xz) == list(zf[0]) and list(z2.yz) == list(zf[1]) and list(z2.xy) == list(zf[2])


@pytest.mark.parametrize("name", ["svm_small.npz", "svm_small_linear.npz", "svm_small_xy.npz", "svm_walabot.npz",
                                  "real_xy_svm.npz", "svm_small_binary.npz"])
def test_svm_oracle_matches_sklearn(name):
    g = load_golden(name)
    m = svm_model_arrays(g)
    if name == "real_xy_svm.npz":
        Xq = g["xy_u8"].reshape(len(g["xy_u8"]), -1)[g["test_idx"]]
    else:
        Xq = g["test_feat_u8"]
    X = Xq.astype(np.float32) / np.float32(255.0)
    C = len(m["classes"])
    dec = O.svm_decision_ovo(X, m["sv"], m["dual_coef"], m["intercept"], m["n_support"], m["gamma"], m["kernel"])
    # float64 round-off only (BLAS dot ordering inside libsvm vs NumPy's)
    if C == 2:       # sklearn flips the sign of the single pair value for binary problems
        np.testing.assert_allclose(-dec[:, 0], g["dec_ovo"], rtol=0, atol=1e-10)
        T = O.sklearn_decision_function(dec, C)
        np.testing.assert_allclose(T, g["dec_ovr"], rtol=0, atol=1e-10)
        np.testing.assert_array_equal(m["classes"][O.svm_vote_labels(dec, C)], g["label_vote"])
        proba = O.calibrated_proba(T, m["calib_a"], m["calib_b"])
        np.testing.assert_allclose(proba, g["proba"], rtol=0, atol=1e-10)
        np.testing.assert_array_equal(m["classes"][O.calibrated_labels(proba)], g["label_calib"])
        return
    np.testing.assert_allclose(dec, g["dec_ovo"], rtol=0, atol=1e-10)
    # W-matrix form (what the GPU epilogue uses) is the same sum
    W = O.ovo_weight_matrix(m["dual_coef"], m["n_support"])
    K = O.svm_kernel_values(X, m["sv"], m["gamma"], m["kernel"])
    np.testing.assert_allclose(K @ W.T + m["intercept"], g["dec_ovo"], rtol=0, atol=1e-10)
    ovr = O.ovr_decision_function(dec, C)
    np.testing.assert_allclose(ovr, g["dec_ovr"], rtol=0, atol=1e-10)
    np.testing.assert_array_equal(m["classes"][O.svm_vote_labels(dec, C)], g["label_vote"])
    proba = O.calibrated_proba(ovr, m["calib_a"], m["calib_b"])
    np.testing.assert_allclose(proba, g["proba"], rtol=0, atol=1e-10)
    np.testing.assert_array_equal(m["classes"][O.calibrated_labels(proba)], g["label_calib"])


def test_max_projection_features_of_fixture_volumes():
    g = load_golden("svm_walabot.npz")
    vol = g["test_vol_u8"].astype(np.float32)
    xz, yz, xy = O.project_max(vol)
    f = O.features_from_projections(xz, yz, xy, tuple(bool(b) for b in g["mask"]), scale=False)
    np.testing.assert_array_equal(f.astype(np.uint8), g["test_feat_u8"])


def test_linear_oracle_matches_sklearn():
    g = load_golden("linear_golden.npz")
    X = g["test_feat_u8"].astype(np.float32) / np.float32(255.0)
    dec = O.linear_decision(X, g["coef"], g["intercept"])
    np.testing.assert_allclose(dec, g["dec"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(g["classes"][np.argmax(dec, axis=1)], g["label"])
    proba = O.calibrated_proba(dec, g["calib_a"], g["calib_b"])
    np.testing.assert_allclose(proba, g["proba"], rtol=0, atol=1e-13)
    np.testing.assert_array_equal(g["classes"][O.calibrated_labels(proba)], g["label_calib"])


def test_classifier_threshold_matches_reference():
    g = load_golden("classifier_threshold.npz")
    names, p = O.classifier_threshold(g["proba"], list(g["class_names"]), 0.7)
    assert names == [str(n) for n in g["names"]]
    np.testing.assert_array_equal(p, g["max_proba"])


def test_projection_identities():
    vol, _ = O.synth_volumes(3, 6, 9, 11, 14)
    xz, yz, xy = O.project_max(vol)
    assert xz.shape == (6, 9, 14) and yz.shape == (6, 11, 14) and xy.shape == (6, 9, 11)
    for b in range(len(vol)):
        a = O.project_max(vol[b])
        np.testing.assert_array_equal(a[0], xz[b]); np.testing.assert_array_equal(a[1], yz[b])
        np.testing.assert_array_equal(a[2], xy[b])
    sx, sy, sz = O.project_sum(vol)
    st, sp, sr = O.axis_energy_profiles(vol[0])
    np.testing.assert_array_equal(st, sx[0].sum(axis=1))
    np.testing.assert_array_equal(sp, sy[0].sum(axis=1))
    np.testing.assert_array_equal(sr, sy[0].sum(axis=0))


def test_libsvm_pairwise_proba_matches_sklearn():
    g = load_golden("svm_platt.npz")
    for tag, C in (("c3", 3), ("c2", 2)):
        dec = g[tag + "_dec"]
        if C == 2:
            dec = -dec               # sklearn flips the sign of the binary pair value; libsvm couples its own
        p = O.libsvm_pairwise_proba(dec, g[tag + "_probA"], g[tag + "_probB"], C)
        np.testing.assert_allclose(p, g[tag + "_proba"], rtol=0, atol=1e-12)


def test_pil_bicubic_resize_matches_pillow():
    """oracle_np.pil_resize_bicubic (restating Pillow's Resample.c) against outputs of Pillow itself for the
    reference's call (dnn.py:202-205, 240-245; sgan.py): bit-exact float32."""
    g = load_golden("pil_resize.npz")
    for name in ("xz", "yz", "xy"):
        for q, want in zip(g["in_" + name], g["out80_" + name]):
            got = O.pil_resize_bicubic(O.scale_unit_range(q), (80, 80))
            assert got.dtype == np.float32
            np.testing.assert_array_equal(got, want)
    for q, want in zip(g["in_xz"], g["out128_xz"]):
        np.testing.assert_array_equal(O.pil_resize_bicubic(O.scale_unit_range(q), (128, 128)), want)
    for k in "abcd":
        want = g["xout_" + k]
        np.testing.assert_array_equal(O.pil_resize_bicubic(g["xin_" + k], want.shape), want)
    # coefficient rows are normalised and the windows stay inside the image
    b, kk = O.pil_resample_coeffs(176, 80)
    assert kk.shape == (80, 11) and np.allclose(kk.sum(1), 1.0) and (b[:, 0] >= 0).all() and (b.sum(1) <= 176).all()


def test_reference_generated_data_fixture():
    """The reference's own sample data (train-results/sgan/generated_data_*.pickle.save, non-integer float32) with the
    rows common.process_samples makes of them and the dnn inputs Pillow makes of them (make_golden.py)."""
    g = load_golden("generated_data.npz")
    samples = [(g["xz"][b], g["yz"][b], g["xy"][b]) for b in range(len(g["xz"]))]
    for sc in (0, 1):
        want = g["feat_scale%d" % sc]
        got = O.process_samples(samples, scale=bool(sc))
        np.testing.assert_array_equal(got, want)
        ident = O.features_from_projections(g["xz"], g["yz"], g["xy"], (True, True, True), bool(sc))
        assert np.abs(ident - want).max() <= (1e-4 if not sc else 1e-6)     # SciPy's spline round trip at zoom 1
    for b in range(2):
        for i, nm in enumerate(("xz", "yz", "xy")):
            got = O.pil_resize_bicubic(O.scale_unit_range(g[nm][b]), (80, 80))
            np.testing.assert_array_equal(got, g["dnn_inputs_80"][b, i])


def test_multiclass_oracle_matches_sklearn():
    """More classes than the reference's three (its label set is open, train.py:656-663): 5-class SVC and 6-class SGD
    golden vectors of tests/golden/make_golden_multiclass.py."""
    g = load_golden("svm_multiclass.npz")
    C = len(g["svc_classes"])
    sv = (g["svc_sv_u8"].astype(np.float32) / np.float32(255.0)).astype(np.float64)
    X = g["svc_test_u8"].astype(np.float32) / np.float32(255.0)
    dec = O.svm_decision_ovo(X, sv, g["svc_dual_coef"], g["svc_intercept"], g["svc_n_support"], float(g["svc_gamma"]), "rbf")
    np.testing.assert_allclose(dec, g["svc_dec_ovo"], rtol=0, atol=1e-10)
    ovr = O.ovr_decision_function(dec, C)
    np.testing.assert_allclose(ovr, g["svc_dec_ovr"], rtol=0, atol=1e-10)
    np.testing.assert_array_equal(g["svc_classes"][O.svm_vote_labels(dec, C)], g["svc_label_vote"])
    proba = O.calibrated_proba(ovr, g["svc_calib_a"], g["svc_calib_b"])
    np.testing.assert_allclose(proba, g["svc_proba"], rtol=0, atol=1e-10)
    np.testing.assert_array_equal(g["svc_classes"][O.calibrated_labels(proba)], g["svc_label_calib"])
    p = O.libsvm_pairwise_proba(dec, g["svc_probA"], g["svc_probB"], C)
    np.testing.assert_allclose(p, g["svc_platt_proba"], rtol=0, atol=1e-12)
    Xs = g["sgd_test_u8"].astype(np.float32) / np.float32(255.0)
    d = O.linear_decision(Xs, g["sgd_coef"], g["sgd_intercept"])
    np.testing.assert_allclose(d, g["sgd_dec"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(g["sgd_classes"][np.argmax(d, axis=1)], g["sgd_label"])
    pr = O.calibrated_proba(d, g["sgd_calib_a"], g["sgd_calib_b"])
    np.testing.assert_allclose(pr, g["sgd_proba"], rtol=0, atol=1e-13)


def test_augmentation_oracle_matches_the_reference_data_generator():
    """oracle_np.aug_* against train.DataGenerator itself (imported from the reference by make_golden.py, draws recorded):
    the same SciPy calls, so bit-identical."""
    g = load_golden("augment_golden.npz")
    labels = list(g["labels"])
    bs = int(g["batch_size"])
    u = list(g["rec_uniform"]); nrm = list(g["rec_normal"])
    planes = [g["in_xz"], g["in_yz"], g["in_xy"]]
    k = 0
    want_y = []
    reps = O.aug_repetitions(labels)
    assert reps == [1, 1, 1, 1, 2, 2, 4]
    for pos in range(0, len(labels), bs):
        for si in range(pos, min(pos + bs, len(labels))):
            for _ in range(reps[si]):
                ang = [u.pop(0) for _ in range(3)]
                for pi, nm in enumerate(("out_xz", "out_yz", "out_xy")):
                    np.testing.assert_array_equal(O.aug_rotate(planes[pi][si], ang[pi]), g[nm][k])
                zf = u.pop(0)
                for pi, nm in enumerate(("out_xz", "out_yz", "out_xy")):
                    np.testing.assert_array_equal(O.aug_clipped_zoom(planes[pi][si], zf), g[nm][k + 1])
                nz = [nrm.pop(0) for _ in range(3)]
                for pi, nm in enumerate(("out_xz", "out_yz", "out_xy")):
                    np.testing.assert_array_equal(O.aug_sparse_noise(planes[pi][si], nz[pi]), g[nm][k + 2])
                want_y += [labels[si]] * 3
                k += 3
    assert k == len(g["out_y"]) and not u and not nrm
    np.testing.assert_array_equal(g["out_y"], want_y)
