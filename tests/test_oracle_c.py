"""The C oracle (oracle/oracle.c) against the golden vectors and the NumPy oracle."""
import numpy as np
import pytest

import oracle_c as OC
import oracle_np as O
from conftest import load_golden, svm_model_arrays


@pytest.mark.parametrize("name", ["svm_small.npz", "svm_small_linear.npz", "svm_small_xy.npz", "svm_walabot.npz",
                                  "svm_small_binary.npz"])
def test_c_oracle_svm_matches_sklearn_golden(name):
    g = load_golden(name)
    m = svm_model_arrays(g)
    X = g["test_feat_u8"].astype(np.float32) / np.float32(255.0)
    out = OC.svm(X, m["sv"], m["dual_coef"], m["intercept"], m["n_support"], m["gamma"], m["kernel"],
                 m["calib_a"], m["calib_b"], threads=2)
    if len(m["classes"]) == 2:
        out["dec_ovo"] = -out["dec_ovo"][:, 0]          # sklearn's binary sign flip
    np.testing.assert_allclose(out["dec_ovo"], g["dec_ovo"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(out["dec_ovr"], g["dec_ovr"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(out["proba"], g["proba"], rtol=0, atol=1e-10)
    np.testing.assert_array_equal(m["classes"][out["label_vote"]], g["label_vote"])
    np.testing.assert_array_equal(m["classes"][out["label_calib"]], g["label_calib"])


def test_c_oracle_projection_and_features():
    g = load_golden("svm_walabot.npz")
    vol = g["test_vol_u8"].astype(np.float32)
    xz, yz, xy = OC.project_max(vol, threads=2)
    for a, b in zip((xz, yz, xy), O.project_max(vol)):
        np.testing.assert_array_equal(a, b)
    f = OC.features(xz, yz, xy, scale=True)
    np.testing.assert_array_equal(f, O.features_from_projections(xz, yz, xy, scale=True))
    np.testing.assert_array_equal(np.rint(f * 255).astype(np.uint8), g["test_feat_u8"])
    c = load_golden("common_golden.npz")
    v = c["volumes_u8"].astype(np.float32)
    sl = OC.project_slice(v, c["slice_ijk"])
    for b in range(len(v)):
        w = O.project_slice(v[b], *c["slice_ijk"][b])
        for pl in range(3):
            np.testing.assert_array_equal(sl[pl][b], w[pl])
    fs = OC.features(*sl, scale=True)
    assert np.abs(fs - c["feat_m0_s1"]).max() <= 1e-15       # the reference's own rows (SciPy zoom at 1.0)
