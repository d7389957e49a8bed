#!/usr/bin/env python3
"""One worker of bench.py's all-cores CPU baseline (tools/bench_support.reference_libs_process_pool): its own process, the
reference's own CPU path (NumPy max -> common.process_samples restated in oracle/oracle_np.py -> scikit-learn
CalibratedClassifierCV(SVC(rbf)).predict) on its slice of the frames.  Frames and model arrays come from .npy files in shared
memory (memory-mapped: one copy for all workers).  Prints one JSON line: frames done, wall-clock start / end of the timed
part, labels.  Not part of the product path."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    d, wid, lo, hi, budget = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass
    import oracle_np as O
    import bench_support as BS
    vh = np.load(os.path.join(d, "frames.npy"), mmap_mode="r")
    sv = np.load(os.path.join(d, "sv_f64.npy"), mmap_mode="r")          # one physical copy for every worker
    model = {k: np.load(os.path.join(d, k + ".npy")) for k in ("dual_coef", "intercept", "n_support", "calib_a", "calib_b", "classes")}
    model["gamma"] = float(np.load(os.path.join(d, "gamma.npy")))
    D = int(vh.shape[1] * vh.shape[3] + vh.shape[2] * vh.shape[3] + vh.shape[1] * vh.shape[2])
    cal = BS.build_sklearn_rbf_model(model, D, sv_f64=sv)
    BS._reference_path(np.asarray(vh[lo:lo + 1]), cal, O)          # touch every code path once (imports, page-ins)
    open(os.path.join(d, "ready_%d" % wid), "w").close()
    go = os.path.join(d, "go")
    while not os.path.exists(go):                                  # common start: the workers' start-up is not the baseline
        time.sleep(0.01)
    t_go = float(open(go).read())
    while time.time() < t_go:
        time.sleep(0.002)
    labels = []
    t0 = time.time()
    pos = lo
    while pos < hi and time.time() - t0 < budget:
        e = min(pos + 4, hi)
        labels.extend(int(v) for v in BS._reference_path(np.asarray(vh[pos:e]), cal, O))
        pos = e
    t1 = time.time()
    print(json.dumps({"lo": lo, "done": pos - lo, "t0": t0, "t1": t1, "labels": labels}), flush=True)


if __name__ == "__main__":
    main()
