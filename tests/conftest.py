import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def built_lib():
    """Path of libradarml_hip.so, (re)built when stale -- hipcc cross-compiles without a GPU."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("rml_build", os.path.join(ROOT, "radar-ml_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build()


@pytest.fixture(scope="session")
def rml(built_lib):
    import radar_ml_amd
    return radar_ml_amd


@pytest.fixture
def rml_opt(rml):
    """rml_opt(name, value): rml_ctx_set_option on the current device's context (radar_ml_amd._lib.OPTIONS names), restored after the test.
    (Rounds 1-5 flipped environment variables the library read with getenv on every launch.)"""
    from radar_ml_amd import _lib
    saved = {}

    def set_(name, value):
        old = _lib.set_option(name, int(value))
        saved.setdefault(name, old)
    yield set_
    for k, v in saved.items():
        _lib.set_option(k, v)


def svm_model_arrays(g):
    """dict of model arrays from a golden svm fixture (SVs back from their uint8 codes)."""
    sv = (g["sv_u8"].astype(np.float32) / np.float32(255.0)).astype(np.float64)
    return dict(sv=sv, dual_coef=g["dual_coef"], intercept=g["intercept"], n_support=g["n_support"],
                gamma=float(g["gamma"]), classes=g["classes"], calib_a=g["calib_a"], calib_b=g["calib_b"],
                kernel=str(g["kernel"]) if "kernel" in g.files else "rbf")
