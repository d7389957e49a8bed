#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/.

Runs ONLY in the build container (needs /root/reference).  It imports the Python
reference (``common.py``, ``predict.py`` with a stub ``WalabotAPI`` module) and
scikit-learn (the reference's third-party SVM implementation) and records
inputs -> outputs as small ``.npz`` fixtures.  Nothing of the reference's source
text is stored -- only data (inputs / expected outputs / data parsed from the
reference's own log files).

    python tests/golden/make_golden.py
"""
import os
import re
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SEED = 1234  # the reference's RANDOM_SEED (train.py:32)


def import_reference():
    stub = types.ModuleType("WalabotAPI")
    stub.PROF_SENSOR = 0
    sys.modules["WalabotAPI"] = stub
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, REF)
    import common  # noqa
    import predict  # noqa
    return common, predict


def synth(seed, n, X, Y, Z):
    sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
    import oracle_np
    return oracle_np.synth_volumes(seed, n, X, Y, Z)


# ----------------------------------------------------------------------------------
def golden_index_kats(common):
    """(x,y,z)->(i,j,k) known answers printed by the reference run:
    ground_truth_samples.log 'x: / y: / z:' blocks followed by 'i: .., j: .., k: ..'."""
    txt = open(os.path.join(REF, "ground_truth_samples.log")).read()
    pat = re.compile(r"\nx: (\S+)\ny: (\S+)\nz: (\S+)\namplitude: \S+\n\n\S+ \S+ __main__\s+DEBUG\s+i: (-?\d+), j: (-?\d+), k: (-?\d+)")
    xyz, ijk = [], []
    for m in pat.finditer(txt):
        xyz.append([float(m.group(1)), float(m.group(2)), float(m.group(3))])
        ijk.append([int(m.group(4)), int(m.group(5)), int(m.group(6))])
    xyz, ijk = np.array(xyz), np.array(ijk, dtype=np.int32)
    # cross-check against the imported reference function
    for (x, y, z), want in zip(xyz, ijk):
        got = common.calculate_matrix_indices(x, y, z, 22, 31, 176)
        assert tuple(got) == tuple(want), (got, want)
    # plus reference-evaluated random targets (incl. out-of-arena -> negative indices)
    rng = np.random.default_rng(SEED)
    rx = rng.uniform(-150, 150, 256); ry = rng.uniform(-120, 120, 256); rz = rng.uniform(5, 400, 256)
    rijk = np.array([common.calculate_matrix_indices(x, y, z, 22, 31, 176)
                     for x, y, z in zip(rx, ry, rz)], dtype=np.int32)
    sph = np.array([common.cartesian_to_spherical(x, y, z) for x, y, z in zip(rx, ry, rz)])
    car = np.array([common.spherical_to_cartesian(r, t, p) for r, t, p in sph])
    np.savez_compressed(os.path.join(HERE, "index_kats.npz"), log_xyz=xyz, log_ijk=ijk,
                        sizes=np.array([22, 31, 176]), rand_xyz=np.stack([rx, ry, rz], 1),
                        rand_ijk=rijk, rand_sph=sph, rand_car=car)
    print("index_kats: %d log KATs, %d random" % (len(xyz), len(rx)))


def golden_common(common, predict):
    """process_samples / get_derived_targets / slices via the imported reference."""
    X, Y, Z = 22, 31, 176
    vol, cls = synth(SEED, 12, X, Y, Z)
    rng = np.random.default_rng(SEED + 1)
    out = {"volumes_u8": vol.astype(np.uint8), "cls": cls}
    # derived targets (common.py:49-80) for 1 and 3 targets
    for nt in (1, 3):
        ijk = []; xyz = []
        for v in vol:
            t = common.DerivedTarget.get_derived_targets(v, X, Y, Z, num_targets=nt)
            ijk.append([[d.i, d.j, d.k] for d in t]); xyz.append([[d.xPosCm, d.yPosCm, d.zPosCm] for d in t])
        out["derived_ijk_%d" % nt] = np.array(ijk, dtype=np.int32)
        out["derived_xyz_%d" % nt] = np.array(xyz)
    # slices at the derived (i,j,k) exactly as predict.py:102-107 / ground_truth_samples.py:413-419
    ijk1 = out["derived_ijk_1"][:, 0, :]
    samples = []
    for v, (i, j, k) in zip(vol, ijk1):
        yz = v[i, :, :]; xz = v[:, j, :]; xy = v[:, :, k]
        samples.append((xz, yz, xy))
    out["slice_ijk"] = ijk1
    # process_samples through the reference for the mask/scale combinations used in the repo
    masks = [(True, True, True), (False, False, True), (True, False, True), (False, True, False)]
    for mi, m in enumerate(masks):
        for sc in (False, True):
            f = common.process_samples(samples, proj_mask=common.ProjMask(*m), scale=sc)
            out["feat_m%d_s%d" % (mi, int(sc))] = f
    out["masks"] = np.array(masks)
    # non-unit zoom through the reference (predict arena != train arena: predict.py:34-54,109-116)
    zoom = predict.calc_proj_zoom(22, 31, 176, 20, 28, 160)
    small = []
    v2, _ = synth(SEED + 2, 4, 20, 28, 160)
    for v in v2:
        t = common.DerivedTarget.get_derived_targets(v, 20, 28, 160)[0]
        small.append((v[:, t.j, :], v[t.i, :, :], v[:, :, t.k]))
    fz = common.process_samples(small, proj_zoom=zoom, scale=True)
    out["zoom_in_xz"] = np.array([s[0] for s in small]); out["zoom_in_yz"] = np.array([s[1] for s in small])
    out["zoom_in_xy"] = np.array([s[2] for s in small])
    out["zoom_factors"] = np.array([zoom.xz, zoom.yz, zoom.xy])
    out["zoom_feat"] = fz
    np.savez_compressed(os.path.join(HERE, "common_golden.npz"), **out)
    print("common_golden: feat shape", out["feat_m0_s1"].shape, "zoom feat", fz.shape)


def fit_svc(Xtr, ytr, Xval, yval, gamma, C=10.0, kernel="rbf"):
    from sklearn import svm
    from sklearn.calibration import CalibratedClassifierCV
    clf = svm.SVC(kernel=kernel, C=C, gamma=gamma, probability=False, class_weight="balanced",
                  random_state=SEED, cache_size=1000)
    clf.fit(Xtr, ytr)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cal = CalibratedClassifierCV(estimator=clf, cv="prefit")   # train.py:723 (kwarg renamed in sklearn>=1.2)
        cal.fit(Xval, yval)
    return clf, cal


def svc_arrays(clf, cal):
    cc = cal.calibrated_classifiers_[0]
    return dict(
        sv_u8=None,
        dual_coef=clf._dual_coef_.copy(), intercept=clf._intercept_.copy(),
        n_support=clf._n_support.astype(np.int32), gamma=np.float64(clf._gamma),
        classes=clf.classes_.copy(),
        calib_a=np.array([c.a_ for c in cc.calibrators]), calib_b=np.array([c.b_ for c in cc.calibrators]),
    )


def golden_svm(common, name, X, Y, Z, ntrain, nval, ntest, gamma, mask=(True, True, True), kernel="rbf", binary=False):
    """Fit the reference's model object (train.py:478-482,723-724 hyper-parameters from
    train-results/train_svc.log:24-31: C=10, gamma=0.01, rbf, class_weight balanced) on
    synthetic max-projection features and record sklearn's outputs."""
    n = ntrain + nval + ntest
    vol, cls = synth(SEED + 10 + X, n, X, Y, Z)
    if binary:      # the reference's optional 'pet' aliasing (dnn.py:36-38 CLASS_ALIAS): person vs pet
        cls = np.minimum(cls, 1).astype(np.int32)
    xz = vol.max(axis=2); yz = vol.max(axis=1); xy = vol.max(axis=3)
    samples = [(a, b, c) for a, b, c in zip(xz, yz, xy)]
    F = common.process_samples(samples, proj_mask=common.ProjMask(*mask), scale=True)
    # process_samples at zoom 1 is identity up to ~1e-16 (spline prefilter round trip);
    # the integer grid k/255 is what the classifier is trained on in the exact path
    Fq = np.rint(F * 255.0).astype(np.uint8)
    Fx = (Fq.astype(np.float32) / np.float32(255.0))
    assert np.abs(F - Fx).max() < 1e-6
    tr, va, te = slice(0, ntrain), slice(ntrain, ntrain + nval), slice(ntrain + nval, n)
    clf, cal = fit_svc(Fx[tr], cls[tr], Fx[va], cls[va], gamma, kernel=kernel)
    arr = svc_arrays(clf, cal)
    sv = clf.support_vectors_
    svq = np.rint(sv * 255.0).astype(np.uint8)
    assert np.array_equal((svq.astype(np.float32) / np.float32(255.0)).astype(np.float64), sv)
    arr["sv_u8"] = svq
    Xte = Fx[te]
    clf.decision_function_shape = "ovo"
    ovo = clf.decision_function(Xte)
    clf.decision_function_shape = "ovr"
    ovr = clf.decision_function(Xte)
    out = dict(arr, grid=np.array([X, Y, Z]), mask=np.array(mask), kernel=np.array(kernel),
               test_feat_u8=Fq[te], test_vol_u8=vol[te].astype(np.uint8), test_cls=cls[te],
               dec_ovo=ovo, dec_ovr=ovr, label_vote=clf.predict(Xte),
               proba=cal.predict_proba(Xte), label_calib=cal.predict(Xte))
    np.savez_compressed(os.path.join(HERE, name), **out)
    acc = (out["label_calib"] == cls[te]).mean()
    kfrac = None
    print("%s: M=%d D=%d ntest=%d acc=%.3f n_support=%s" % (name, sv.shape[0], sv.shape[1], ntest, acc,
                                                           arr["n_support"]))


def golden_platt(common):
    """SVC(probability=True) as the reference constructs it (train.py:478): libsvm's own Platt scaling +
    pairwise coupling (SVC.predict_proba), recorded next to the pair decision values it is computed from."""
    from sklearn import svm
    X, Y, Z = 8, 10, 16
    vol, cls = synth(SEED + 31, 700, X, Y, Z)
    F = np.concatenate([vol.max(axis=2).reshape(700, -1), vol.max(axis=1).reshape(700, -1),
                        vol.max(axis=3).reshape(700, -1)], axis=1)
    Fq = F.astype(np.uint8); Fx = Fq.astype(np.float32) / np.float32(255.0)
    out = {}
    for tag, y in (("c3", cls), ("c2", np.minimum(cls, 1))):
        clf = svm.SVC(kernel="rbf", C=10.0, gamma=0.05, probability=True, class_weight="balanced", random_state=SEED)
        clf.fit(Fx[:500], y[:500])
        te = slice(500, 700)
        clf.decision_function_shape = "ovo"
        out.update({tag + "_sv_u8": np.rint(clf.support_vectors_ * 255).astype(np.uint8), tag + "_dual_coef": clf._dual_coef_,
                    tag + "_intercept": clf._intercept_, tag + "_n_support": clf._n_support.astype(np.int32),
                    tag + "_classes": clf.classes_, tag + "_probA": clf._probA, tag + "_probB": clf._probB,
                    tag + "_dec": clf.decision_function(Fx[te]), tag + "_proba": clf.predict_proba(Fx[te]),
                    tag + "_label_vote": clf.predict(Fx[te])})
    out["test_feat_u8"] = Fq[500:700]; out["gamma"] = np.float64(0.05)
    np.savez_compressed(os.path.join(HERE, "svm_platt.npz"), **out)
    print("svm_platt: M3=%d M2=%d" % (out["c3_sv_u8"].shape[0], out["c2_sv_u8"].shape[0]))


def golden_real_xy():
    """The only real radar data in the reference tree: the DEBUG data dump in
    ground_truth_samples.log (XY projections printed in full; XZ/YZ elided)."""
    txt = open(os.path.join(REF, "ground_truth_samples.log")).read()
    start = txt.index("Data dump:")
    body = txt[start:]
    lab_start = body.index("'labels':")
    arrays = body[:lab_start].split("array(")[1:]
    xy = []
    for a in arrays:
        if "..." in a:
            continue
        nums = re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", a.split("dtype")[0])
        v = np.array([float(t) for t in nums], dtype=np.float32)
        assert v.size == 22 * 31, v.size
        xy.append(v.reshape(22, 31))
    labels = re.findall(r"'(\w+)'", body[lab_start:].split("[", 1)[1].split("]")[0])
    xy = np.array(xy)
    assert len(labels) == len(xy), (len(labels), len(xy))
    assert np.array_equal(xy, np.rint(xy)) and xy.min() >= 0 and xy.max() <= 255
    names = sorted(set(labels))
    y = np.array([names.index(l) for l in labels], dtype=np.int32)
    Fq = xy.reshape(len(xy), -1).astype(np.uint8)
    Fx = Fq.astype(np.float32) / np.float32(255.0)
    rng = np.random.default_rng(SEED)
    perm = rng.permutation(len(xy))
    tr, va, te = perm[:340], perm[340:415], perm[415:]
    clf, cal = fit_svc(Fx[tr], y[tr], Fx[va], y[va], gamma=0.01)
    arr = svc_arrays(clf, cal)
    arr["sv_u8"] = np.rint(clf.support_vectors_ * 255.0).astype(np.uint8)
    clf.decision_function_shape = "ovo"; ovo = clf.decision_function(Fx[te])
    clf.decision_function_shape = "ovr"; ovr = clf.decision_function(Fx[te])
    np.savez_compressed(os.path.join(HERE, "real_xy_svm.npz"), **dict(
        arr, xy_u8=xy.astype(np.uint8), labels=y, label_names=np.array(names), test_idx=te,
        dec_ovo=ovo, dec_ovr=ovr, label_vote=clf.predict(Fx[te]), proba=cal.predict_proba(Fx[te]),
        label_calib=cal.predict(Fx[te]), mask=np.array([False, False, True]), grid=np.array([22, 31, 176])))
    print("real_xy_svm: %d samples %s, M=%d, acc=%.3f" % (len(xy), np.bincount(y), arr["sv_u8"].shape[0],
                                                          (cal.predict(Fx[te]) == y[te]).mean()))


def golden_linear(common):
    """SGD (logistic) classifier, the reference's default model (train.py:350-381, 421, 433)."""
    from sklearn import linear_model
    from sklearn.calibration import CalibratedClassifierCV
    X, Y, Z = 10, 12, 16
    vol, cls = synth(SEED + 77, 700, X, Y, Z)
    F = np.concatenate([vol.max(axis=2).reshape(700, -1), vol.max(axis=1).reshape(700, -1),
                        vol.max(axis=3).reshape(700, -1)], axis=1)
    # scikit-learn 0.24 (requirements.txt:57) runs SGD in float64 only; newer versions keep float32
    # inputs in float32, so hand it the float32 values widened to float64 to pin 0.24's arithmetic.
    Fq = F.astype(np.uint8); Fx = (Fq.astype(np.float32) / np.float32(255.0)).astype(np.float64)
    clf = linear_model.SGDClassifier(loss="log_loss", alpha=1e-4, max_iter=200, tol=1e-4,
                                     class_weight="balanced", random_state=SEED)
    clf.fit(Fx[:500], cls[:500])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cal = CalibratedClassifierCV(estimator=clf, cv="prefit").fit(Fx[500:600], cls[500:600])
    cc = cal.calibrated_classifiers_[0]
    te = slice(600, 700)
    np.savez_compressed(os.path.join(HERE, "linear_golden.npz"), coef=clf.coef_, intercept=clf.intercept_,
                        classes=clf.classes_, calib_a=np.array([c.a_ for c in cc.calibrators]),
                        calib_b=np.array([c.b_ for c in cc.calibrators]), test_feat_u8=Fq[te],
                        dec=clf.decision_function(Fx[te]), label=clf.predict(Fx[te]),
                        proba=cal.predict_proba(Fx[te]), label_calib=cal.predict(Fx[te]))
    print("linear_golden: coef", clf.coef_.shape, "acc", (clf.predict(Fx[te]) == cls[te]).mean())


def golden_classifier_threshold(predict):
    """predict.classifier (predict.py:56-70) on a fake model returning fixed probabilities."""
    rng = np.random.default_rng(SEED)
    P = rng.dirichlet([0.6, 0.6, 0.6], size=64)

    class LE:  # label encoder stand-in
        classes_ = np.array(["cat", "dog", "person"])

    names, probs = [], []
    for row in P:
        class M:
            def predict_proba(self, obs, row=row):
                return row[None, :]
        n, p = predict.classifier(np.zeros(4), M(), LE(), min_proba=0.7)
        names.append(str(n)); probs.append(p)
    np.savez_compressed(os.path.join(HERE, "classifier_threshold.npz"), proba=P, names=np.array(names),
                        max_proba=np.array(probs), class_names=LE.classes_)
    print("classifier_threshold:", sum(n == "Unknown" for n in names), "Unknown of", len(names))


def golden_generated_data(common):
    """The only sample data the reference ships in its own data-set format (datasets/README.md:8-20):
    train-results/sgan/generated_data_{0230,1035}.pickle.save -- GAN output, i.e. NON-integer float32 projections of
    the Walabot arena grid.  A few samples of each with what the reference computes from them: common.process_samples
    (all masks on, scale on/off; common.py:123-149) and the dnn preprocessing (dnn.py:202-205, 240-245 via Pillow)."""
    import pickle
    from PIL import Image
    xs, tags = [], []
    for name in ("generated_data_0230", "generated_data_1035"):
        with open(os.path.join(REF, "train-results", "sgan", name + ".pickle.save"), "rb") as fp:
            d = pickle.load(fp)
        assert set(d.keys()) == {"samples", "labels"} and len(d["samples"]) == 100
        for idx in (0, 57):
            xs.append(tuple(np.asarray(p) for p in d["samples"][idx]))
            tags.append("%s[%d] label=%s" % (name, idx, d["labels"][idx]))
    out = {"tags": np.array(tags)}
    for i, nm in enumerate(("xz", "yz", "xy")):
        out[nm] = np.stack([s[i] for s in xs])
        assert out[nm].dtype == np.float32
    pm = common.ProjMask(xz=True, yz=True, xy=True)
    out["feat_scale0"] = common.process_samples(xs, proj_mask=pm, scale=False)
    out["feat_scale1"] = common.process_samples(xs, proj_mask=pm, scale=True)
    res = []
    for s in xs[:2]:
        res.append(np.stack([np.asarray(Image.fromarray((p - 255.0 / 2.) / (255.0 / 2.)).resize((80, 80), resample=Image.BICUBIC))
                             for p in s]))
    out["dnn_inputs_80"] = np.stack(res)                 # (2, 3, 80, 80): xz, yz, xy of the first two samples
    np.savez_compressed(os.path.join(HERE, "generated_data.npz"), **out)
    print("generated_data:", {k: v.shape for k, v in out.items()})


def golden_augment(common):
    """train.DataGenerator (train.py:32-185) imported from the reference and run on a small imbalanced set of Walabot-grid
    projections, balance=True, with its two random sources seeded (np.random.seed for the np.random.uniform draws,
    np.random.PCG64 replaced by a seeded one for the `rg.normal` draws) and every draw RECORDED, so that the kernels can be
    checked on the recorded parameters and the Python mirror on the seeds."""
    os.environ.setdefault("MPLBACKEND", "Agg")
    import train                                            # the reference module (sklearn / scipy / matplotlib only)
    vol, cls = synth(SEED + 91, 7, 22, 31, 176)
    samples = []
    for v in vol:
        xz, yz, xy = v.max(axis=1), v.max(axis=0), v.max(axis=2)
        samples.append(tuple((p / np.float32(255.0)).astype(np.float32) for p in (xz, yz, xy)))     # train.py:667 scaling
    samples[3] = tuple(p * np.float32(0.37) for p in samples[3])                                   # non-grid values as well
    labels = [0, 0, 0, 0, 1, 1, 2]
    rec = {"uniform": [], "normal": []}
    real_uniform, RealPCG = np.random.uniform, np.random.PCG64

    def uniform(lo, hi):
        v = real_uniform(lo, hi); rec["uniform"].append(v); return v

    class RecGen(np.random.Generator):
        def normal(self, loc=0.0, scale=1.0, size=None):
            v = super().normal(loc, scale, size); rec["normal"].append(float(v)); return v

    np.random.seed(SEED)
    np.random.uniform = uniform
    np.random.PCG64 = lambda: RealPCG(SEED + 1)
    RealGen = np.random.Generator
    np.random.Generator = RecGen
    try:
        gen = train.DataGenerator(rotation_range=15.0, zoom_range=0.3, noise_sd=0.2, balance=True)
        flow = gen.flow(list(samples), list(labels), batch_size=4)
        b1x, b1y = next(flow)
        b2x, b2y = next(flow)
    finally:
        np.random.uniform, np.random.PCG64, np.random.Generator = real_uniform, RealPCG, RealGen
    aug = list(b1x) + list(b2x)
    out = dict(in_xz=np.stack([s[0] for s in samples]), in_yz=np.stack([s[1] for s in samples]), in_xy=np.stack([s[2] for s in samples]),
               labels=np.array(labels), batch_size=np.int64(4), seed_uniform=np.int64(SEED), seed_pcg=np.int64(SEED + 1),
               out_xz=np.stack([t[0] for t in aug]), out_yz=np.stack([t[1] for t in aug]), out_xy=np.stack([t[2] for t in aug]),
               out_y=np.concatenate([b1y, b2y]), n_batch1=np.int64(len(b1y)),
               rec_uniform=np.array(rec["uniform"]), rec_normal=np.array(rec["normal"]))
    np.savez_compressed(os.path.join(HERE, "augment_golden.npz"), **out)
    print("augment_golden: %d augmented tuples from %d samples, %d uniform / %d normal draws, out dtype %s"
          % (len(aug), len(samples), len(rec["uniform"]), len(rec["normal"]), aug[0][0].dtype))


def golden_pil_resize():
    """The resize in front of the dnn / sgan classifiers exactly as the reference calls it (dnn.py:202-205, 240-245;
    sgan.py:638-641, 676-681): scale to [-1,1], Image.fromarray(p).resize(RESCALE, resample=Image.BICUBIC).  Inputs
    are uint8-valued projections of the Walabot arena grid (22, 31, 176); outputs come from Pillow itself."""
    from PIL import Image
    import PIL
    rng = np.random.default_rng(77)
    out = {"pillow_version": np.array(PIL.__version__)}
    X, Y, Z = 22, 31, 176
    shapes = {"xz": (X, Z), "yz": (Y, Z), "xy": (X, Y)}
    for name, (h, w) in shapes.items():
        p = rng.integers(0, 256, size=(2, h, w)).astype(np.uint8)
        p[0] = (np.add.outer(np.arange(h), np.arange(w)) * 255 // (h + w - 2)).astype(np.uint8)     # a smooth ramp
        out["in_" + name] = p
        for tag, rescale in (("80", (80, 80)), ("128", (128, 128))):         # dnn.py:33 / sgan.py RESCALE
            if tag == "128" and name != "xz":
                continue
            res = []
            for q in p:
                d = (q - 255.0 / 2.) / (255.0 / 2.)
                res.append(np.asarray(Image.fromarray(d).resize(rescale, resample=Image.BICUBIC)))
            out["out%s_%s" % (tag, name)] = np.array(res)
            assert out["out%s_%s" % (tag, name)].dtype == np.float32
    # other geometries: pure downscale, pure upscale, one axis unchanged, non-square target (width, height)
    extra = {"a": ((100, 37), (48, 64)), "b": ((5, 300), (80, 80)), "c": ((80, 31), (80, 80)), "d": ((9, 9), (64, 96))}
    for k, ((h, w), (oh, ow)) in extra.items():
        q = rng.uniform(-1, 1, size=(h, w)).astype(np.float32)
        out["xin_" + k] = q
        out["xout_" + k] = np.asarray(Image.fromarray(q).resize((ow, oh), resample=Image.BICUBIC))
        assert out["xout_" + k].shape == (oh, ow)
    np.savez_compressed(os.path.join(HERE, "pil_resize.npz"), **out)
    print("pil_resize:", {k: v.shape for k, v in out.items() if k.startswith("out")})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pil":      # Pillow-only fixture: does not need /root/reference
        golden_pil_resize()
        sys.exit(0)
    common, predict = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "generated":
        golden_generated_data(common)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "augment":
        golden_augment(common)
        sys.exit(0)
    golden_index_kats(common)
    golden_common(common, predict)
    golden_svm(common, "svm_small.npz", 8, 10, 16, 500, 120, 256, gamma=0.05)
    golden_svm(common, "svm_small_linear.npz", 8, 10, 16, 400, 100, 128, gamma=0.05, kernel="linear")
    golden_svm(common, "svm_small_xy.npz", 8, 10, 16, 400, 100, 128, gamma=0.05, mask=(False, False, True))
    golden_svm(common, "svm_walabot.npz", 22, 31, 176, 420, 100, 128, gamma=0.01)
    golden_svm(common, "svm_small_binary.npz", 8, 10, 16, 400, 100, 128, gamma=0.05, binary=True)
    golden_platt(common)
    golden_real_xy()
    golden_linear(common)
    golden_classifier_threshold(predict)
    golden_pil_resize()
    golden_generated_data(common)
    golden_augment(common)
