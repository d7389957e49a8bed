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
            return {"value": pool["value"], "unit": "frames/s", "cores": pool["cores"], "kind": "port", "port_of": "the reference's own library calls (NumPy max, scipy.ndimage.zoom, scikit-learn CalibratedClassifierCV(SVC).predict), restated call for call",
                    "sample": "%d of the same synthetic frames on %d worker processes (one per core) in %.1f s: %s" % (pool["frames"], pool["cores"], pool["seconds"], what),
                    "label_mismatch_vs_gpu": pool["label_mismatch_vs_gpu"], "note": note,
                    "single_process": single, "thread_pool": thread_pool, "process_pool": pool_row}
        return {"value": thread_pool["value"], "unit": "frames/s", "cores": int(threads), "kind": "port", "port_of": "the reference's own library calls (NumPy max, scipy.ndimage.zoom, scikit-learn CalibratedClassifierCV(SVC).predict), restated call for call",
                "sample": "%d of the same synthetic frames on %d threads of one process in %.1f s: %s" % (done, threads, dt2, what),
                "label_mismatch_vs_gpu": mism2, "note": note,
                "single_process": single, "thread_pool": thread_pool, "process_pool": pool_row}
    return {"value": thread_pool["value"], "unit": "frames/s", "cores": int(threads), "kind": "port", "port_of": "the reference's own library calls (NumPy max, scipy.ndimage.zoom, scikit-learn CalibratedClassifierCV(SVC).predict), restated call for call",
            "sample": "%d of the same synthetic frames on %d threads in %.1f s (the process pool could not be run): %s" % (done, threads, dt2, what),
            "label_mismatch_vs_gpu": mism2, "note": note, "single_process": single, "thread_pool": thread_pool}


def reference_libs_process_pool(vh, model, gpu_label_idx, workers, budget_s=12.0, ready_timeout_s=90.0):
    """All host cores the honest way: one PROCESS per core (NumPy ``max`` and ``ndimage.zoom`` hold the GIL, so a thread pool
    measures the interpreter lock, not the machine), each running the reference path on its own slice of the frames.
    Frames and the float64 SV matrix sit once in /dev/shm (memory-mapped by the workers: one copy for all of them); the workers
    import their libraries and build the scikit-learn object first, report ready, and start together when the parent says
    go; the rate is (frames done by all) / (latest end - earliest start).  Returns the object or None when the pool could
    not be run."""
    import tempfile
    classes = np.asarray(model["classes"])
    n = int(len(vh))
    workers = int(max(1, min(workers, n // 4 if n >= 4 else 1)))
    d = tempfile.mkdtemp(prefix="rml_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
    procs = []
    try:
        np.save(os.path.join(d, "frames.npy"), np.ascontiguousarray(vh))
        np.save(os.path.join(d, "sv_f64.npy"), (np.asarray(model["sv_u8"]).astype(np.float32) / np.float32(255.0)).astype(np.float64))
        for k in ("dual_coef", "intercept", "n_support", "calib_a", "calib_b", "classes"):
            np.save(os.path.join(d, k + ".npy"), np.ascontiguousarray(model[k]))
        np.save(os.path.join(d, "gamma.npy"), np.float64(model["gamma"]))
        per = (n + workers - 1) // workers
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        for w in range(workers):
            lo, hi = w * per, min(n, (w + 1) * per)
            if lo >= hi:
                break
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "cpu_worker.py"), d, str(w), str(lo), str(hi), str(budget_s)],
                                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, cwd="/tmp"))
        t_wait = time.time()
        while sum(os.path.exists(os.path.join(d, "ready_%d" % w)) for w in range(len(procs))) < len(procs):
            if time.time() - t_wait > ready_timeout_s or any(p.poll() not in (None, 0) for p in procs):
                return None
            time.sleep(0.05)
        with open(os.path.join(d, "go.tmp"), "w") as f:
            f.write(repr(time.time() + 0.5))
        os.replace(os.path.join(d, "go.tmp"), os.path.join(d, "go"))
        outs = []
        deadline = time.time() + budget_s + 60.0
        for p in procs:
            try:
                o, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
            except subprocess.TimeoutExpired:
                return None
            if p.returncode != 0:
                return None
            outs.append(json.loads(o.decode().strip().splitlines()[-1]))
    except Exception:
        return None
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(d, ignore_errors=True)
    done = sum(o["done"] for o in outs)
    span = max(o["t1"] for o in outs) - min(o["t0"] for o in outs)
    mism = 0
    for o in outs:
        lab = np.asarray(o["labels"], dtype=np.int64)
        mism += int((lab != classes[gpu_label_idx[o["lo"]:o["lo"] + len(lab)]]).sum())
    return {"value": round(done / span, 2) if span > 0 else 0.0, "unit": "frames/s", "cores": len(outs), "frames": int(done),
            "seconds": round(span, 2), "start_skew_s": round(max(o["t0"] for o in outs) - min(o["t0"] for o in outs), 3),
            "label_mismatch_vs_gpu": mism}


# ------------------------------------------------------------------------------------------------------------------
# HBM traffic of the projection launches from this run's own counters
# ------------------------------------------------------------------------------------------------------------------
def _find_db(d):
    hits = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True) + glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    return hits[0] if hits else None


def _read_counter(db, counter):
    """[(kernel_name, value, duration_ns)] in dispatch order"""
    c = sqlite3.connect(db)
    cur = c.execute("select * from counters_collection limit 1")
    cols = [d[0] for d in cur.description]
    order = next((k for k in ("dispatch_id", "start", "start_timestamp", "timestamp", "id") if k in cols), None)
    q = "select kernel_name, value, duration from counters_collection where counter_name=?" + (" order by %s" % order if order else "")
    return c.execute(q, (counter,)).fetchall()


def measure_traffic(configs, timeout_s=240):
    """configs: [{"tag", "grid": [X,Y,Z], "frames", "u8": bool}].  Returns {tag: {"hbm_bytes", "fetch_bytes",
    "write_bytes", "kernel", "dispatches"}} or None.  One child process per counter runs every configuration."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    child = os.path.join(ROOT, "tools", "pmc_child.py")
    spec = json.dumps(configs)
    got = {}
    tmp = tempfile.mkdtemp(prefix="rml_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", RML_WAVE_SHARE="1")      # the child launches the kernel configuration of the fused pipeline
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "k", "--", sys.executable, child, spec]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
            except Exception:
                return None
            if r.returncode != 0:
                return None
            order = None
            for line in r.stdout.decode(errors="replace").splitlines():
                if line.startswith("PMC_CHILD_ORDER "):
                    order = json.loads(line[len("PMC_CHILD_ORDER "):])
            db = _find_db(out)
            if db is None or order is None:
                return None
            rows = [x for x in _read_counter(db, counter) if any(k in x[0] for k in ("k_project", "k_derive_slice", "k_slice_rows"))]
            # the child launches every configuration `reps` times in order and nothing else that is called k_project*
            reps = order["reps"]
            if len(rows) != reps * len(order["tags"]):
                return None
            for i, tag in enumerate(order["tags"]):
                mine = rows[i * reps:(i + 1) * reps][1:]          # first launch of a configuration = warm-up
                got.setdefault(tag, {})[counter] = statistics.mean(v for _, v, _ in mine)
                got[tag]["kernel"] = mine[0][0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
                got[tag]["dispatches"] = len(mine)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for tag, g in got.items():
        if "FETCH_SIZE" not in g or "WRITE_SIZE" not in g:
            return None
        fetch = 2.0 * g["FETCH_SIZE"] * 1024.0      # KB as reported; gfx950 tallies 128-B requests at 64 B (guide, HBM section)
        write = g["WRITE_SIZE"] * 1024.0
        res[tag] = {"hbm_bytes": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "kernel": g["kernel"],
                    "dispatches": g["dispatches"],
                    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, this run), 2x FETCH correction"}
    return res
