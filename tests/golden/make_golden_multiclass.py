#!/usr/bin/env python3
"""Golden vectors for models with MORE than the reference's three classes (its label set is open, train.py:656-663):
a 5-class RBF SVC + CalibratedClassifierCV(prefit) + libsvm Platt/pairwise coupling, and a 6-class SGD classifier,
fitted and evaluated by scikit-learn in the build container (same pinning category as make_golden.py's SVM files:
the SVM arithmetic lives in scikit-learn, the reference only calls it -- train.py:478-482, 722-724).

    python tests/golden/make_golden_multiclass.py        # writes svm_multiclass.npz next to this file
"""
import os
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 1234


def rows(rng, n, ncls, D):
    """integer-code feature rows (0..255) around ncls sparse class templates -> (codes uint8, labels)"""
    tmpl = rng.integers(0, 256, (ncls, D)) * (rng.random((ncls, D)) < 0.25)
    y = rng.integers(0, ncls, n)
    F = 0.45 * tmpl[y] + rng.normal(40, 70, (n, D)) * (rng.random((n, D)) < 0.6)
    return np.clip(np.rint(F), 0, 255).astype(np.uint8), y.astype(np.int64)


def main():
    from sklearn import svm, linear_model
    from sklearn.calibration import CalibratedClassifierCV
    rng = np.random.default_rng(SEED)
    out = {}
    # ---- 5-class RBF SVC, probability=True as train.py:478 constructs it ----
    D, ncls = 296, 5
    Fq, y = rows(rng, 900, ncls, D)
    Fx = Fq.astype(np.float32) / np.float32(255.0)              # train.py:667
    tr, va, te = slice(0, 600), slice(600, 750), slice(750, 900)
    clf = svm.SVC(kernel="rbf", C=10.0, gamma=0.05, probability=True, class_weight="balanced", random_state=SEED)
    clf.fit(Fx[tr], y[tr])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cal = CalibratedClassifierCV(estimator=clf, cv="prefit").fit(Fx[va], y[va])
    cc = cal.calibrated_classifiers_[0]
    clf.decision_function_shape = "ovo"; ovo = clf.decision_function(Fx[te])
    clf.decision_function_shape = "ovr"; ovr = clf.decision_function(Fx[te])
    out.update(svc_sv_u8=np.rint(clf.support_vectors_ * 255).astype(np.uint8), svc_dual_coef=clf._dual_coef_,
               svc_intercept=clf._intercept_, svc_n_support=clf._n_support.astype(np.int32), svc_classes=clf.classes_,
               svc_gamma=np.float64(0.05), svc_probA=clf._probA, svc_probB=clf._probB,
               svc_calib_a=np.array([c.a_ for c in cc.calibrators]), svc_calib_b=np.array([c.b_ for c in cc.calibrators]),
               svc_test_u8=Fq[te], svc_dec_ovo=ovo, svc_dec_ovr=ovr, svc_label_vote=clf.predict(Fx[te]),
               svc_platt_proba=clf.predict_proba(Fx[te]), svc_proba=cal.predict_proba(Fx[te]), svc_label_calib=cal.predict(Fx[te]))
    print("svc: %d classes, M=%d, acc=%.3f" % (ncls, clf.support_vectors_.shape[0], (cal.predict(Fx[te]) == y[te]).mean()))
    # ---- 6-class SGD (logistic) classifier, float64 as scikit-learn 0.24 runs it ----
    D, ncls = 240, 6
    Fq, y = rows(rng, 900, ncls, D)
    Fx = (Fq.astype(np.float32) / np.float32(255.0)).astype(np.float64)
    sgd = linear_model.SGDClassifier(loss="log_loss", alpha=1e-4, max_iter=300, tol=1e-4, class_weight="balanced", random_state=SEED)
    sgd.fit(Fx[tr], y[tr])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cal = CalibratedClassifierCV(estimator=sgd, cv="prefit").fit(Fx[va], y[va])
    cc = cal.calibrated_classifiers_[0]
    out.update(sgd_coef=sgd.coef_, sgd_intercept=sgd.intercept_, sgd_classes=sgd.classes_,
               sgd_calib_a=np.array([c.a_ for c in cc.calibrators]), sgd_calib_b=np.array([c.b_ for c in cc.calibrators]),
               sgd_test_u8=Fq[te], sgd_dec=sgd.decision_function(Fx[te]), sgd_label=sgd.predict(Fx[te]),
               sgd_proba=cal.predict_proba(Fx[te]), sgd_label_calib=cal.predict(Fx[te]))
    print("sgd: %d classes, acc=%.3f" % (ncls, (cal.predict(Fx[te]) == y[te]).mean()))
    np.savez_compressed(os.path.join(HERE, "svm_multiclass.npz"), **out)


if __name__ == "__main__":
    main()
