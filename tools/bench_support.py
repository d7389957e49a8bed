"""Measurement helpers of bench.py that are NOT part of the product path:

* ``reference_libs_baseline``: the reference's own CPU path with the reference's own libraries -- NumPy ``max`` per axis
  -> ``common.process_samples`` (``scipy.ndimage.zoom(p, 1.0)`` + concatenate + ``/255``: common.py:141-148, restated in
  oracle/oracle_np.py) -> scikit-learn ``CalibratedClassifierCV(SVC(kernel='rbf')).predict`` (train.py:217,723-724;
  predict.py:60) -- timed single-process as the reference runs it and on all host cores through a thread pool
  (libsvm's predict releases the GIL).  SURVEY.md §8d.
* ``measure_traffic``: HBM bytes per projection launch from this run's own PMC counters (``rocprofv3 --pmc FETCH_SIZE`` and
  ``--pmc WRITE_SIZE`` in two separate kernel-trace-only passes over tools/pmc_child.py, 2x FETCH correction for gfx950 --
  /opt/skills/guides/MI355X_MICROARCH.md, HBM section).  Returns None when rocprofv3 is not usable.

Only bench.py imports this module; it may use oracle/ (the product may not).
"""
import concurrent.futures
import glob
import json
import os
import shutil
import sqlite3
import statistics
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------------------------
# reference-library CPU baseline
# ------------------------------------------------------------------------------------------------------------------
def build_sklearn_rbf_model(model, D, sv_f64=None):
    """The object the reference pickles -- CalibratedClassifierCV(prefit, sigmoid) around SVC(kernel='rbf') -- carrying the
    bench model's arrays.  (bench.py fits with kernel='precomputed' on the GPU Gram matrix; libsvm's predict needs an RBF
    estimator, so a tiny RBF SVC is fitted for its private state and the fitted arrays are then replaced.)"""
    import copy
    import warnings
    from sklearn import svm
    from sklearn.calibration import CalibratedClassifierCV
    classes = np.asarray(model["classes"])
    C = len(classes)
    # train.py:667 scaling; ``sv_f64``: the same matrix already made (the pool's workers share one memory-mapped copy)
    sv = sv_f64 if sv_f64 is not None else (model["sv_u8"].astype(np.float32) / np.float32(255.0)).astype(np.float64)
    M = sv.shape[0]
    rng = np.random.default_rng(0)
    Xd = rng.random((4 * C, D))
    yd = np.repeat(classes, 4)
    svc = svm.SVC(kernel="rbf", C=10.0, gamma=float(model["gamma"]), class_weight="balanced")
    svc.fit(Xd, yd)
    svc.support_vectors_ = sv if (isinstance(sv, np.ndarray) and sv.flags["C_CONTIGUOUS"] and sv.dtype == np.float64) else np.ascontiguousarray(sv, dtype=np.float64)
    svc.support_ = np.arange(M, dtype=np.int32)
    svc._n_support = np.asarray(model["n_support"], dtype=np.int32)
    svc._dual_coef_ = np.ascontiguousarray(model["dual_coef"], dtype=np.float64)
    svc.dual_coef_ = svc._dual_coef_
    svc._intercept_ = np.ascontiguousarray(model["intercept"], dtype=np.float64)
    svc.intercept_ = svc._intercept_
    svc._gamma = float(model["gamma"])
    svc._probA = np.empty(0, dtype=np.float64)
    svc._probB = np.empty(0, dtype=np.float64)
    svc.shape_fit_ = (M, D)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cal = CalibratedClassifierCV(estimator=svc, cv="prefit").fit(Xd, yd)
    cc = cal.calibrated_classifiers_[0]
    for c, (a, b) in enumerate(zip(model["calib_a"], model["calib_b"])):
        cc.calibrators[c].a_ = float(a)
        cc.calibrators[c].b_ = float(b)
    return cal


def _reference_path(frames, cal, O):
    """frames (n,X,Y,Z) float32 -> labels, exactly the reference's sequence of library calls"""
    samples = [(v.max(axis=1), v.max(axis=0), v.max(axis=2)) for v in frames]          # (xz, yz, xy), common.py:40
    feats = O.process_samples(samples, scale=True)                                      # ndimage.zoom(p, 1.0) + concatenate + /255
    return cal.predict(feats)


def reference_libs_baseline(vh, model, gpu_label_idx, threads, budget_s=25.0, want_all=2048):
    """Time the reference path on the host cores.  vh: host frames the GPU also classified; gpu_label_idx: their calibrated
    label indices from the GPU.  Returns the ``cpu_baseline`` object of the bench line."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_np as O
    try:
        from threadpoolctl import threadpool_limits, threadpool_info
    except Exception:                               # pragma: no cover
        threadpool_limits = None
    D = int(sum(a * b for a, b in ((vh.shape[1], vh.shape[3]), (vh.shape[2], vh.shape[3]), (vh.shape[1], vh.shape[2]))))
    cal = build_sklearn_rbf_model(model, D)
    classes = np.asarray(model["classes"])
    ctx = threadpool_limits(limits=1) if threadpool_limits else None       # BLAS/OpenMP pools pinned to 1: libsvm itself is serial
    try:
        # (i) single process, as the reference runs it
        t0 = time.perf_counter()
        lab = [_reference_path(vh[:4], cal, O)]
        probe = (time.perf_counter() - t0) / 4
        n1 = int(max(8, min(len(vh), (budget_s * 0.4) / max(probe, 1e-6))))
        t0 = time.perf_counter()
        lab1 = _reference_path(vh[:n1], cal, O)
        dt1 = time.perf_counter() - t0
        mism1 = int((lab1 != classes[gpu_label_idx[:n1]]).sum())
        # (ii) all host cores: frames chunked over a thread pool (libsvm releases the GIL; NumPy max and SciPy zoom mostly hold it)
        nall = int(min(len(vh), want_all))
        chunk = 4
        jobs = [(s, min(s + chunk, nall)) for s in range(0, nall, chunk)]
        done = 0
        mism2 = 0
        t0 = time.perf_counter()
        with concurrent.futures.ThreadPoolExecutor(max_workers=threads) as ex:
            futs = {}
            it = iter(jobs)
            # keep the pool full, stop feeding it once the time budget is spent
            for _ in range(threads * 2):
                j = next(it, None)
                if j is None:
                    break
                futs[ex.submit(_reference_path, vh[j[0]:j[1]], cal, O)] = j
            while futs:
                fin, _ = concurrent.futures.wait(futs, return_when=concurrent.futures.FIRST_COMPLETED)
                for f in fin:
                    s, e = futs.pop(f)
                    mism2 += int((f.result() != classes[gpu_label_idx[s:e]]).sum())
                    done += e - s
                    if time.perf_counter() - t0 < budget_s * 0.6:
                        j = next(it, None)
                        if j is not None:
                            futs[ex.submit(_reference_path, vh[j[0]:j[1]], cal, O)] = j
        dt2 = time.perf_counter() - t0
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    import sklearn, scipy
    what = ("numpy max -> scipy.ndimage.zoom(p, 1.0) + concatenate + /255 (common.process_samples) -> sklearn %s "
            "CalibratedClassifierCV(SVC(rbf)).predict; scipy %s; BLAS/OpenMP pools limited to 1 thread" % (sklearn.__version__, scipy.__version__))
    thread_pool = {"value": round(done / dt2, 2), "unit": "frames/s", "cores": int(threads), "frames": int(done), "seconds": round(dt2, 1),
                   "label_mismatch_vs_gpu": mism2, "note": "one process, thread pool over 4-frame chunks: GIL-limited (numpy max and ndimage.zoom hold it)"}
    single = {"value": round(n1 / dt1, 2), "unit": "frames/s", "cores": 1, "frames": n1, "seconds": round(dt1, 1), "label_mismatch_vs_gpu": mism1}
    # (iii) all host cores as processes: what the node's cores give the reference path when the interpreter lock is out of the way
    pool = reference_libs_process_pool(vh[:nall], model, gpu_label_idx[:nall], threads, budget_s=budget_s * 0.5)
    # libsvm's predict streams the whole float64 support-vector matrix once per ROW (8*M*D bytes: 420 MB at the headline model), so
    # every all-core variant is bound by the host's memory system, not by its cores; the headline value is the best of the two
    # all-core runs, both are listed
    sv_mb = round(8.0 * float(model["sv_u8"].shape[0]) * float(model["sv_u8"].shape[1]) / 1e6, 1)
    note = ("libsvm streams the float64 support-vector matrix%s once per row: all-core figures are memory-bound; "
            "value = the better of the process pool and the thread pool" % ((" (%.0f MB)" % sv_mb) if sv_mb else ""))
    if pool is not None and pool["value"] > 0:
        pool_row = {"value": pool["value"], "unit": "frames/s", "cores": pool["cores"], "frames": pool["frames"], "seconds": pool["seconds"],
                    "label_mismatch_vs_gpu": pool["label_mismatch_vs_gpu"], "start_skew_s": pool["start_skew_s"],
                    "note": "one worker process per core, the model's arrays shared through page-cache-backed memory maps"}
        if pool["value"] >= thread_pool["value"]:
            return {"value": pool["value"], "unit": "frames/s", "cores": pool["cores"], "kind": "reference-libs",
                    "sample": "%d of the same synthetic frames on %d worker processes (one per core) in %.1f s: %s" % (pool["frames"], pool["cores"], pool["seconds"], what),
                    "label_mismatch_vs_gpu": pool["label_mismatch_vs_gpu"], "note": note,
                    "single_process": single, "thread_pool": thread_pool, "process_pool": pool_row}
        return {"value": thread_pool["value"], "unit": "frames/s", "cores": int(threads), "kind": "reference-libs",
                "sample": "%d of the same synthetic frames on %d threads of one process in %.1f s: %s" % (done, threads, dt2, what),
                "label_mismatch_vs_gpu": mism2, "note": note,
                "single_process": single, "thread_pool": thread_pool, "process_pool": pool_row}
    return {"value": thread_pool["value"], "unit": "frames/s", "cores": int(threads), "kind": "reference-libs",
            "sample": "%d of the same synthetic frames on %d threads in %.1f s (the process pool could not be run): %s" % (done, threads, dt2, what),
            "label_mismatch_vs_gpu": mism2, "note": note, "single_process": single, "thread_pool": thread_pool}
