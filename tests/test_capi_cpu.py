"""CPU-side checks: the C-ABI library loads and exports every symbol include/radarml.h declares,
fails cleanly without a GPU, and the host-side mirror of the reference interface behaves like the
reference (names, defaults, errors).  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "radarml.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rml_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libradarml_hip.so does not export %s" % s


def test_python_binding_covers_header(rml):
    from radar_ml_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()
    lib = _lib.load()
    assert lib.rml_version().decode().startswith("radarml-hip")
    assert lib.rml_feature_len(22, 31, 176, 7) == 10010          # train_svc.log:19
    assert lib.rml_feature_len(22, 31, 176, 4) == 682
    assert lib.rml_feature_len(64, 64, 128, 7) == 20480


def test_ctx_create_fails_cleanly_without_gpu(rml):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from radar_ml_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.rml_ctx_create(0, ctypes.byref(h))
    assert rc < 0 and not h.value
    assert lib.rml_last_error()
    # product path fails loudly, no CPU fallback
    with pytest.raises(rml.RadarMLError):
        rml.process_samples([(np.zeros((2, 4), np.float32), np.zeros((3, 4), np.float32), np.zeros((2, 3), np.float32))])
    with pytest.raises(rml.RadarMLError):
        rml.project(np.zeros((1, 2, 3, 4), np.float32))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "radar-ml_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle|oracle_np|oracle/", src, flags=re.M), \
                    "%s reaches into oracle/" % f


def test_host_mirror_names_and_defaults(rml):
    assert rml.ProjMask._fields == ("xz", "yz", "xy") and rml.ProjZoom._fields == ("xz", "yz", "xy")
    assert rml.RADAR_MAX == 255.0 and rml.RADAR_MIN == 0.0
    import inspect
    sig = inspect.signature(rml.process_samples)
    assert list(sig.parameters) == ["samples", "proj_mask", "proj_zoom", "scale"]
    assert sig.parameters["proj_mask"].default == rml.ProjMask(True, True, True)
    assert sig.parameters["proj_zoom"].default == rml.ProjZoom([1.0, 1.0], [1.0, 1.0], [1.0, 1.0])
    assert sig.parameters["scale"].default is False
    assert rml.DerivedTarget._fields == ("xPosCm", "yPosCm", "zPosCm", "amplitude", "i", "j", "k")


def test_calculate_matrix_indices_known_answers(rml):
    g = load_golden("index_kats.npz")
    X, Y, Z = (int(v) for v in g["sizes"])
    for (x, y, z), want in zip(g["log_xyz"], g["log_ijk"]):
        got = rml.calculate_matrix_indices(x, y, z, X, Y, Z)
        assert got == tuple(want) and all(isinstance(v, int) for v in got)
    # batched form (incl. out-of-arena targets -> negative / too large indices, not clamped)
    xyz = g["rand_xyz"]
    got = rml.calculate_matrix_indices(xyz[:, 0], xyz[:, 1], xyz[:, 2], X, Y, Z)
    np.testing.assert_array_equal(got, g["rand_ijk"])
    sph = np.array(rml.cartesian_to_spherical(xyz[:, 0], xyz[:, 1], xyz[:, 2])).T
    np.testing.assert_array_equal(sph, g["rand_sph"])
    car = np.array(rml.spherical_to_cartesian(sph[:, 0], sph[:, 1], sph[:, 2])).T
    np.testing.assert_array_equal(car, g["rand_car"])


def test_calc_proj_zoom_and_classifier(rml):
    g = load_golden("common_golden.npz")
    z = rml.calc_proj_zoom(22, 31, 176, 20, 28, 160)
    zf = g["zoom_factors"]
    assert list(z.xz) == list(zf[0]) and list(z.yz) == list(zf[1]) and list(z.xy) == list(zf[2])
    assert rml.calc_proj_zoom(22, 31, 176, 22, 31, 176) == rml.ProjZoom([1.0, 1.0], [1.0, 1.0], [1.0, 1.0])
    t = load_golden("classifier_threshold.npz")

    class LE:
        classes_ = t["class_names"]

    for row, name, p in zip(t["proba"], t["names"], t["max_proba"]):
        class M:
            def predict_proba(self, obs, row=row):
                assert obs.shape == (1, 4)
                return row[None, :]
        n, pr = rml.classifier(np.zeros(4), M(), LE(), min_proba=0.7)
        assert str(n) == str(name) and pr == p
    names, p = rml.classify_batch(t["proba"], list(t["class_names"]), 0.7)
    assert [str(n) for n in names] == [str(n) for n in t["names"]]


def test_process_samples_argument_errors(rml):
    a = np.zeros((2, 4), np.float32); b = np.zeros((3, 4), np.float32); c = np.zeros((2, 3), np.float32)
    with pytest.raises(ValueError):      # ragged, like np.array() in the reference
        rml.process_samples([(a, b, c), (np.zeros((2, 5), np.float32), b, c)])
    with pytest.raises(ValueError):
        rml.process_samples([(a, b, c)], proj_mask=rml.ProjMask(False, False, False))
    assert rml.process_samples([]).shape == (0,)


def test_dataset_pickle_roundtrip(rml, tmp_path):
    """datasets/README.md:8-20 format: write, append (ground_truth_samples.py:561-587), read, alias, filter."""
    import importlib
    ds = importlib.import_module("radar_ml_amd.datasets")
    rng = np.random.default_rng(0)
    xz = rng.integers(0, 255, (5, 22, 176)).astype(np.float32)
    yz = rng.integers(0, 255, (5, 31, 176)).astype(np.float32)
    xy = rng.integers(0, 255, (5, 22, 31)).astype(np.float32)
    p = str(tmp_path / "radar_samples.pickle")
    assert ds.save_dataset(p, xz[:3], yz[:3], xy[:3], ["person", "polly", "rebel"]) == 3
    assert ds.save_dataset(p, xz[3:], yz[3:], xy[3:], ["dog", "person"]) == 5
    import pickle
    d = pickle.load(open(p, "rb"))
    assert set(d) == {"samples", "labels"} and len(d["samples"]) == 5 and len(d["samples"][0]) == 3
    a, b, c, labels = ds.load_dataset(p, label_alias={"polly": "dog", "rebel": "cat"})
    np.testing.assert_array_equal(a, xz); np.testing.assert_array_equal(b, yz); np.testing.assert_array_equal(c, xy)
    assert labels == ["person", "dog", "cat", "dog", "person"]
    a, b, c, labels = ds.load_dataset([p, p], desired_labels={"person"})
    assert a.shape == (4, 22, 176) and labels == ["person"] * 4


def test_unwrap_calibrated_accepts_sklearn_024_attribute_names():
    """The reference pins scikit-learn 0.24 (requirements.txt:57): its pickled _CalibratedClassifier carries
    ``base_estimator`` / ``calibrators_`` where 1.x has ``estimator`` / ``calibrators``."""
    import types
    import importlib
    svm = importlib.import_module("radar_ml_amd.svm")
    cals = [types.SimpleNamespace(a_=-1.5 - i, b_=0.25 * i) for i in range(3)]
    est = types.SimpleNamespace(name="est")
    new = types.SimpleNamespace(estimator=est, calibrators=cals, method="sigmoid")
    old = types.SimpleNamespace(base_estimator=est, calibrators_=cals, method="sigmoid")
    for cc in (new, old):
        e, a, b = svm.unwrap_calibrated(cc)
        assert e is est
        np.testing.assert_array_equal(a, [-1.5, -2.5, -3.5])
        np.testing.assert_array_equal(b, [0.0, 0.25, 0.5])
    with pytest.raises(NotImplementedError):
        svm.unwrap_calibrated(types.SimpleNamespace(estimator=est, calibrators=cals, method="isotonic"))
    with pytest.raises(NotImplementedError):
        svm.unwrap_calibrated(types.SimpleNamespace(foo=1))
    with pytest.raises(NotImplementedError):
        svm._check_classes(svm.MAX_CLASSES + 1)


def test_context_creation_failure_raises_instead_of_deadlocking(monkeypatch):
    """_lib.context() used to call check() -> load() while holding a non-reentrant lock: a failing rml_ctx_create hung the
    process.  Without a GPU rml_ctx_create fails, which is exactly the case to exercise."""
    import threading
    import importlib
    import torch
    _lib = importlib.import_module("radar_ml_amd._lib")
    if torch.cuda.is_available():
        pytest.skip("needs a box without a HIP device")
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    res = {}

    def run():
        try:
            _lib.context(0)
            res["r"] = "no error"
        except _lib.RadarMLError as e:
            res["r"] = str(e)

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(20)
    assert not t.is_alive(), "context() dead-locked on a failing rml_ctx_create"
    assert "rml_ctx_create failed" in res["r"]


def test_slice_indices_accept_every_integer_dtype():
    """common._slice_indices: the range check runs on the caller's integers widened to int64 -- uint8 / int8 / uint16 / int64 index
    arrays are accepted when in range (a narrow dtype must not wrap the bound: -64 as uint8 is 192; int8 cannot hold Z = 128),
    out-of-range and wrapped values raise IndexError like NumPy indexing would."""
    import torch
    from radar_ml_amd.common import _slice_indices
    cpu = torch.device("cpu")
    X, Y, Z = 64, 64, 128
    base = np.array([[0, 1, 2], [63, 63, 127], [5, 6, 100]])
    for dt in (np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64):
        t, T = _slice_indices(base.astype(dt), 3, X, Y, Z, cpu)
        assert T == 1 and t.dtype == torch.int32 and t.tolist() == base.tolist()
    t, _ = _slice_indices(np.array([[-64, -1, 100], [3, 4, 5]], dtype=np.int8), 2, X, Y, Z, cpu)     # int8 cannot hold 128
    assert t.tolist() == [[-64, -1, 100], [3, 4, 5]]
    t, _ = _slice_indices(torch.tensor([[1, 2, 3]], dtype=torch.uint8), 1, X, Y, Z, cpu)
    assert t.tolist() == [[1, 2, 3]]
    for bad in (np.array([[64, 0, 0]], np.uint8), np.array([[0, 0, 128]], np.int64), np.array([[0, -65, 0]], np.int16),
                np.array([[0, 0, 2 ** 32 + 5]], np.int64), np.array([[0, 0, 2 ** 63 + 5]], np.uint64), np.array([[200, 0, 0]], np.uint8)):
        with pytest.raises(IndexError):
            _slice_indices(bad, 1, X, Y, Z, cpu)
    with pytest.raises(IndexError):
        _slice_indices(np.array([[0.0, 1.0, 2.0]]), 1, X, Y, Z, cpu)


def test_host_side_under_address_and_ub_sanitizers():
    """SURVEY.md 5 (sanitizer target): the HOST half of every csrc/*.hip compiled with -fsanitize=address,undefined
    (tools/sanitize/build.py: hipcc --cuda-host-only, seconds) and driven through the C ABI and the host-only internals --
    argument validation of every front door, thread-local error messages from four threads, the model packing of rml_svm_load on
    edge shapes (on / off the code grid, NaN, 2..6 classes, linear kernel), Pillow's table builder, the shape predicates, and the
    graceful failure of rml_ctx_create on a box without a GPU.  A sanitizer report aborts the driver; its own checks exit 1."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools", "sanitize"))
    import importlib
    sb = importlib.import_module("build")
    sys.path.pop(0)
    lib, drv = sb.build()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([drv], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, env=env, cwd="/tmp")
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-4000:]
    assert "phase A: ok" in out and "ALL OK" in out and "runtime error" not in out and "AddressSanitizer" not in out, out[-4000:]
