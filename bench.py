#!/usr/bin/env python3
"""Benchmark of the radar-ml hot path on MI355X: radar frames/s, 3-D volume -> max-projection ->
RBF-SVM label (BASELINE.json metric; workload = configs[2]: projection + RBF-SVM decision function,
3 classes, ~2k support vectors, batch 65 536 frames of 64x64x128 resident in HBM per GPU).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the rank's resident batch (frames shard across ranks,
weak scaling, no data-path collective; the per-frame labels are all-gathered over RCCL at the end
of every step, inside the timed region).  Rank 0 prints the verbose rows as {"doc": ...} on one line and then,
LAST, the contract line (one JSON object < 4 KB: metric, value, roofline, cpu_baseline, summary).
`python bench.py --gpus N` without a launcher starts its own N ranks.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec HBM3E
I8_MFMA_PEAK_TOPS = 3944.0        # MI355X_MICROARCH.md: dense int8 MFMA (>= 3944 TOPS)


def gemm_roofline(lib, ctx):
    """In-situ time of the GEMM + finish kernels of every chunk (second stream) -> achieved int8 TOP/s."""
    nl, ms, ops = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
    lib.rml_profile_read_gemm(ctx, ctypes.byref(nl), ctypes.byref(ms), ctypes.byref(ops))
    if nl.value == 0 or ms.value <= 0:
        return None
    ach = ops.value / (ms.value * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "k_svm_gemm<I8> + k_svm_finish", "achieved": round(ach, 1), "peak": I8_MFMA_PEAK_TOPS,
            "unit": "TOP/s", "frac": round(ach / I8_MFMA_PEAK_TOPS, 4), "launches": int(nl.value),
            "avg_chunk_ms": round(ms.value / nl.value, 4), "note": "algorithmic 2*D*M ops per frame; runs concurrently with the projection of the next chunk"}


def _bench_support():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_support
    return bench_support


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default: WORLD_SIZE of the launcher, else 1")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=65536, help="frames per GPU (configs[2]: 65536)")
    ap.add_argument("--grid", default="64x64x128", help="XxYxZ; 22x31x176 = Walabot arena grid")
    ap.add_argument("--train", type=int, default=3400, help="synthetic frames used to fit the SVC (CPU plumbing)")
    ap.add_argument("--gamma", type=float, default=0.01)
    ap.add_argument("--parity", type=int, default=4096, help="frames checked against the CPU oracle")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU baseline sample (0 = auto ~15 s)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-walabot", action="store_true", help="skip the secondary Walabot-arena-grid workload")
    ap.add_argument("--no-u8", action="store_true", help="skip the uint8-ingest row (the same frames as 1-byte voxels)")
    ap.add_argument("--no-slice", action="store_true", help="skip the slice-projection rows (derive -> slice -> SVM, slice with given ijk)")
    ap.add_argument("--ingest", choices=["f32", "u8"], default="f32",
                    help="u8: the headline step itself runs on uint8 volumes (per-workload profiles: tools/profile_round.sh); "
                         "the line's value is then the uint8 rate")
    ap.add_argument("--walabot-frames", type=int, default=262144, help="frames per GPU of the 22x31x176 workload")
    ap.add_argument("--no-dnn", action="store_true", help="skip the multi-view CNN inference row (BASELINE configs[3])")
    ap.add_argument("--dnn-frames", type=int, default=65536, help="frames per GPU of the CNN inference row")
    ap.add_argument("--no-sgan", action="store_true", help="skip the SGAN discriminator train-step row (configs[4])")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC passes that measure roofline.traffic (it is null then)")
    ap.add_argument("--no-general", action="store_true", help="skip the general_rows row (non-integer data, float64 MFMA path)")
    ap.add_argument("--general-frames", type=int, default=16384, help="frames per GPU of the general_rows row")
    ap.add_argument("--dnn-parity", type=int, default=1024, help="frames of the CNN row checked against the NumPy restatement")
    ap.add_argument("--dnn-train-steps", type=int, default=600, help="float32 Adam steps that give the CNN row's model real margins")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--doc-file", default=None, help="also write the verbose rows (and the contract line) to this JSON file")
    return ap.parse_args()


def fit_model(rml, torch, X, Y, Z, ntrain, gamma, seed, dev):
    """CPU plumbing (BASELINE config 1 / SURVEY §8d): fit the reference's model object -- SVC(rbf, C=10,
    class_weight='balanced') + CalibratedClassifierCV(prefit, sigmoid) (train.py:478-482,722-724) -- on
    synthetic max-projection features.  The Gram matrix comes from the library's own kernel-matrix service
    (rml_svm_kernel_matrix: the exact-integer MFMA path) and is handed to scikit-learn's SMO as a precomputed
    kernel: same optimisation problem, minutes faster than libsvm's own kernel evaluations at D = 20 480."""
    import warnings
    from sklearn import svm
    from sklearn.calibration import CalibratedClassifierCV
    nval = max(300, ntrain // 8)
    v, cls = rml.synth_volumes(ntrain + nval, X, Y, Z, seed=seed, frame0=1 << 40)
    feat = rml.process_volumes(v, mode="max", scale=False)              # integer codes 0..255, float32
    rows = rml.process_volumes(v, mode="max", scale=True)               # train.py:667: float32(code / 255.), IEEE division
    del v
    km = rml.KernelMatrix(rows[:ntrain].cpu().numpy(), gamma)           # Gram-matrix service (csrc/svm.hip, exact path)
    assert km.exact
    K = km.against(rows).cpu().numpy()                                  # (ntrain + nval, ntrain) float64
    del km, rows
    y = cls.cpu().numpy()
    tr, va = slice(0, ntrain), slice(ntrain, ntrain + nval)
    clf = svm.SVC(kernel="precomputed", C=10.0, class_weight="balanced", cache_size=2000)
    clf.fit(K[tr, tr], y[tr])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cal = CalibratedClassifierCV(estimator=clf, cv="prefit").fit(K[va, tr], y[va])
    cc = cal.calibrated_classifiers_[0]
    sup = clf.support_
    codes = feat[tr][torch.as_tensor(sup, device=feat.device, dtype=torch.long)].cpu().numpy().astype(np.uint8)
    kfrac = float((K[tr, tr] > 1e-6).mean())
    model = dict(
        sv_u8=codes, dual_coef=clf._dual_coef_.copy(), intercept=clf._intercept_.copy(),
        n_support=clf._n_support.astype(np.int32), gamma=float(gamma), classes=clf.classes_.copy(),
        calib_a=np.array([c.a_ for c in cc.calibrators]), calib_b=np.array([c.b_ for c in cc.calibrators]),
        val_acc=float((cal.predict(K[va, tr]) == y[va]).mean()), kfrac=kfrac)
    return model


def sv_f64(model):
    # train.py:667 scaling: float32(code / 255), widened to float64 as libsvm sees it
    return (model["sv_u8"].astype(np.float32) / np.float32(255.0)).astype(np.float64)


class Window:
    """callable: the next window of ``src`` (B frames, ``slide`` frames further on each call, ``nslide`` positions);
    .home() = the window [0, B)"""
    def __init__(self, src, B, slide, nslide):
        self.src, self.B, self.slide, self.nslide, self.k = src, B, slide, nslide, 0

    def __call__(self):
        self.k = (self.k + 1) % self.nslide
        return self.src[self.k * self.slide:self.k * self.slide + self.B]

    def home(self):
        return self.src[:self.B]


def run_workload(a, env, grid, frames, primary):
    """Fit the model, make the resident batch, time K steps, and (rank 0) check parity / CPU baseline.
    Returns the result dict on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    rml, _lib, dev, rank, world = env["rml"], env["lib_mod"], env["dev"], env["rank"], env["world"]
    X, Y, Z = grid
    D = rml.feature_len(X, Y, Z)
    frame_bytes = 4 * X * Y * Z

    # ---- model: rank 0 fits, everyone gets the same arrays ---------------------------------
    t0 = time.time()
    obj = [None]
    if rank == 0:
        obj[0] = fit_model(rml, torch, X, Y, Z, a.train, a.gamma, a.seed, dev)
    if world > 1:
        dist.broadcast_object_list(obj, src=0)
    model = obj[0]
    fit_s = time.time() - t0
    svc = rml.GpuSVC(sv_f64(model), model["dual_coef"], model["intercept"], model["n_support"], model["gamma"],
                     model["classes"], calib_a=model["calib_a"], calib_b=model["calib_b"])
    assert svc.exact, "synthetic radar codes must be on the integer grid"
    M = int(model["sv_u8"].shape[0])
    torch.cuda.empty_cache()

    # ---- resident batch ---------------------------------------------------------------------
    free, total = torch.cuda.mem_get_info(dev)
    reserve = 6 << 30
    # Rows whose code stores are read-compare-write (uint8 ingest, derive -> slice: rml_code_rmw) step over a SLIDING window of the
    # resident frames: step k classifies frames [129 k, 129 k + B), so that no row of a chunk workspace ever meets the codes the
    # same frame left there one step earlier (129 is incommensurate with every chunk size) -- what the read-back saves is then what
    # it saves on fresh frames.  The headline float32 step stores plainly and keeps the fixed window [0, B).
    SLIDE = 129
    nslide = a.steps + a.warmup + 3
    B = int(min(frames, max(128, (free - reserve) // frame_bytes - SLIDE * nslide)))
    Vall, cls = rml.synth_volumes(B + SLIDE * nslide, X, Y, Z, seed=a.seed, frame0=rank * frames, device=dev)
    if a.ingest == "u8":
        Vall = Vall.to(torch.uint8)
        frame_bytes = X * Y * Z
    V = Vall[:B]

    def with_rmw(flag, fn):
        """fn() with RML_OPT_CODE_RMW forced on this rank's context -- the with / without pair of a read-compare-write row"""
        with _lib.options(dev, code_rmw=int(flag)):
            return fn()
    lib = _lib.load()
    ctx = _lib.context(dev)
    from radar_ml_amd import dist as rdist

    def step():
        out = svc.decide_volumes(V, mode="max", scale=True, want_proba=True)
        if world > 1:
            out["all_labels"] = rdist.gather_labels(out["label_calib"])      # RCCL all-gather, 4 B/frame
        return out

    for _ in range(a.warmup):
        out = step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    lib.rml_profile_enable(ctx, 1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    nl, ms, nf = ctypes.c_int64(), ctypes.c_double(), ctypes.c_int64()
    lib.rml_profile_read(ctx, ctypes.byref(nl), ctypes.byref(ms), ctypes.byref(nf))
    groof = gemm_roofline(lib, ctx)
    lib.rml_profile_enable(ctx, 0)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # ---- the same frames as uint8 volumes (the radar's native magnitudes): a separate
    #      roofline row, 1 byte per voxel.  Same model, same step, labels must be identical. -----------------
    u8 = None
    if not a.no_u8 and a.ingest != "u8":
        V8all = Vall.to(torch.uint8)
        win8 = Window(V8all, B, SLIDE, nslide)

        def step8(vol=None):
            o = svc.decide_volumes(win8() if vol is None else vol, mode="max", scale=True, want_proba=True)
            if world > 1:
                o["all_labels"] = rdist.gather_labels(o["label_calib"])
            return o

        def timed8():
            step8()
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            lib.rml_profile_enable(ctx, 1)
            t0_ = time.perf_counter()
            for _ in range(a.steps):
                step8()
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0_

        # the pair without the read-compare-write first (its profile counters are discarded), then the row itself
        dt8_plain = with_rmw("0", timed8)
        lib.rml_profile_enable(ctx, 0)
        dt8 = timed8()
        nl8, ms8, nf8 = ctypes.c_int64(), ctypes.c_double(), ctypes.c_int64()
        lib.rml_profile_read(ctx, ctypes.byref(nl8), ctypes.byref(ms8), ctypes.byref(nf8))
        groof8 = gemm_roofline(lib, ctx)
        lib.rml_profile_enable(ctx, 0)
        t8 = torch.tensor([dt8], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t8, op=dist.ReduceOp.MAX)
        dt8 = float(t8.item())
        # the two ingests may run on different exact-GEMM tiles (128x128 / 256x256): the int32 dot products are the same integers
        # and every kernel writes one float64 partial per 128 SV rows, summed alike -- the outputs must be the same bits
        t8p = torch.tensor([dt8_plain], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t8p, op=dist.ReduceOp.MAX)
        dt8_plain = float(t8p.item())
        out8 = step8(win8.home())                       # untimed: the frames of the float32 step, for the identity checks
        lab_same = bool(torch.equal(out8["label_calib"], out["label_calib"]) and torch.equal(out8["label_vote"], out["label_vote"]))
        dec_diff = float((out8["dec_ovo"] - out["dec_ovo"]).abs().max())
        same = lab_same and bool(torch.equal(out8["dec_ovo"], out["dec_ovo"])) and bool(torch.equal(out8["proba"], out["proba"]))
        l8 = max(1, nl8.value)
        a8 = ms8.value / l8
        ach8 = (X * Y * Z + 16) * (nf8.value / l8) / (a8 * 1e-3) / 1e9 if a8 > 0 else 0.0
        tr8 = None          # filled in by main() from this run's own PMC passes (tools/bench_support.measure_traffic)
        u8 = {"value": round(world * B * a.steps / dt8, 1), "unit": "frames/s", "ms_per_step": round(dt8 / a.steps * 1e3, 3),
              "workload": "the same %d frames/GPU as uint8 volumes (1 byte per voxel); every step takes a window %d frames further on" % (B, SLIDE),
              "read_compare_write": {"on_by_default": bool(lib.rml_code_rmw_default(D, X * Y * Z, 0, 1)),
                                     "value_with_plain_stores": round(world * B * a.steps / dt8_plain, 1),
                                     "gain": round(dt8_plain / dt8 - 1.0, 4),
                                     "note": "same steps with RML_OPT_CODE_RMW = 0; sliding windows: no row meets its own frame's old codes"},
              "identical_to_f32_ingest": same, "labels_identical": lab_same, "dec_ovo_max_abs_diff_vs_f32_ingest": dec_diff,
              "hbm_frac_end_to_end": round(B * a.steps / dt8 * (X * Y * Z + 16) / 1e9 / HBM_PEAK_GBS, 4),
              "roofline": {"bound": "hbm", "kernel": "k_project_u8_max" if Z % 16 == 0 else "k_project_fast<uint8>",
                           "achieved": round(ach8, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(ach8 / HBM_PEAK_GBS, 4), "traffic": tr8, "launches": int(l8),
                           "avg_launch_ms": round(a8, 4), "frames_per_launch": nf8.value / l8,
                           "algorithmic_bytes_per_frame": X * Y * Z + 16},
              "gemm_roofline": groof8}
        del V8all, win8, out8
        torch.cuda.empty_cache()

    # ---- the reference-faithful projection: plane SLICES through a target voxel
    #      (predict.py:98-107), with the voxel derived on the GPU (common.py:49-80) or given -- separate roofline rows ----
    slice_rows = None
    if not a.no_slice and a.ingest != "u8":
        slice_rows = {}
        ijk_dev = rml.derive_targets(V, 1)[:, 0, :].contiguous()           # for the "given" row: what an SDK would report

        winS = Window(Vall, B, SLIDE, nslide)

        def timed_steps(fn):
            fn()
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            lib.rml_profile_enable(ctx, 1)
            t0_ = time.perf_counter()
            for _ in range(a.steps):
                fn()
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            dts = time.perf_counter() - t0_
            o = None
            nl_, ms_, nf_ = ctypes.c_int64(), ctypes.c_double(), ctypes.c_int64()
            lib.rml_profile_read(ctx, ctypes.byref(nl_), ctypes.byref(ms_), ctypes.byref(nf_))
            gr = gemm_roofline(lib, ctx)
            lib.rml_profile_enable(ctx, 0)
            tt = torch.tensor([dts], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return o, float(tt.item()), max(1, nl_.value), ms_.value, nf_.value, gr

        ijk_all = rml.derive_targets(Vall, 1)[:, 0, :].contiguous()

        def slice_given():
            v = winS()
            return svc.decide_volumes(v, mode="slice", ijk=ijk_all[winS.k * SLIDE:winS.k * SLIDE + B], scale=True, want_proba=True, validate_ijk=False)

        for key, fn, home, kern, alg in (
                ("derive_slice_svm", lambda: svc.decide_volumes(winS(), mode="slice", scale=True, want_proba=True),
                 lambda: svc.decide_volumes(V, mode="slice", scale=True, want_proba=True), "k_derive_slice", frame_bytes + 16),
                ("slice_mode", slice_given,
                 lambda: svc.decide_volumes(V, mode="slice", ijk=ijk_dev, scale=True, want_proba=True, validate_ijk=False),
                 "k_slice_rows", 4 * D + 12 + 16)):
            plain = None
            if key == "derive_slice_svm":
                plain = with_rmw("0", lambda: timed_steps(fn))[1]
            _, dt_s, nl_s, ms_s, nf_s, gr_s = timed_steps(fn)
            o_s = home()                                # untimed: the window [0, B), for the parity checks below
            avg_s = ms_s / nl_s
            ach_s = alg * (nf_s / nl_s) / (avg_s * 1e-3) / 1e9 if avg_s > 0 else 0.0
            val_s = world * B * a.steps / dt_s
            slice_rows[key] = {"value": round(val_s, 1), "unit": "frames/s", "ms_per_step": round(dt_s / a.steps * 1e3, 3),
                               "hbm_frac_end_to_end": round(val_s / world * alg / 1e9 / HBM_PEAK_GBS, 4),
                               "roofline": {"bound": "hbm", "kernel": kern, "achieved": round(ach_s, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": round(ach_s / HBM_PEAK_GBS, 4), "traffic": None, "launches": int(nl_s),
                                            "avg_launch_ms": round(avg_s, 4), "frames_per_launch": nf_s / nl_s,
                                            "algorithmic_bytes_per_frame": alg},
                               "gemm_roofline": gr_s, "_out": o_s}
            if plain is not None:
                slice_rows[key]["read_compare_write"] = {"on_by_default": bool(lib.rml_code_rmw_default(D, frame_bytes, 1, 0)),
                                                         "value_with_plain_stores": round(world * B * a.steps / plain, 1),
                                                         "gain": round(plain / dt_s - 1.0, 4),
                                                         "note": "same steps with RML_OPT_CODE_RMW = 0; every step takes a window %d frames further on" % SLIDE}
        if "ijk" in slice_rows["derive_slice_svm"]["_out"]:
            slice_rows["derive_slice_svm"]["ijk_equal_rml_derive_targets"] = bool(
                torch.equal(slice_rows["derive_slice_svm"]["_out"]["ijk"], ijk_dev))
        slice_rows["slice_mode"]["request_floor_bytes_per_frame"] = 4 * (X * Z + Y * Z) + 128 * X * Y + D + 16
        slice_rows["slice_mode"]["note"] = ("algorithmic = 4*D read + 28 B; every xy value lives in its own 128-byte memory request (rows are "
                                            ">= 512 B apart), so no kernel fetches less than request_floor_bytes_per_frame")
        # what this row is bound by: not its projection (a gather of 4*D bytes per frame) but the int8 GEMM behind it -- quoted as such,
        # with the projection's in-situ rate against the request floor (what the memory system must move) beside the algorithmic one
        sm = slice_rows["slice_mode"]
        floor_b = sm["request_floor_bytes_per_frame"]
        rf_s = sm["roofline"]
        ach_floor = floor_b * rf_s["frames_per_launch"] / (rf_s["avg_launch_ms"] * 1e-3) / 1e9 if rf_s["avg_launch_ms"] > 0 else 0.0
        rf_s["frac_of_request_floor"] = round(ach_floor / HBM_PEAK_GBS, 4)
        sm["bound"] = {"by": "mfma", "kernel": (sm.get("gemm_roofline") or {}).get("kernel"), "frac": (sm.get("gemm_roofline") or {}).get("frac"),
                       "note": "the exact int8 GEMM of the rows is this row's time; the gather runs beside it at frac_of_request_floor of 8 TB/s"}

    if rank != 0:
        del V, Vall, out, svc
        torch.cuda.empty_cache()
        return None

    # ---- roofline of the dominant kernel (projection; HBM-bound) ---------------------------
    # algorithmic bytes per frame of the fused path: the volume read + 16 B of outputs
    alg_bytes_frame = frame_bytes + 16
    launches = max(1, nl.value)
    avg_ms = ms.value / launches
    frames_per_launch = nf.value / launches
    achieved = alg_bytes_frame * frames_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None          # filled in by main() from this run's own PMC passes (tools/bench_support.measure_traffic)
    zq = Z // 4
    rpl = 1 if 32 < zq <= 64 else (64 // zq if zq in (16, 32) and Y % (64 // zq) == 0 else 0)
    wave = _lib.get_option("waveframe", dev) != 0 and rpl > 0 and Y // rpl <= 32      # wave_kernel_wanted(share_cu) of csrc/project.hip
    kname = "k_project_wave" if wave else ("k_project_fast" if zq & (zq - 1) == 0 else "k_project_rowgroup")
    if wave and ((zq == 44 and 16 < Y <= 32) or (zq in (40, 48, 56) and 8 < Y <= 32)) and _lib.get_option("linplane", dev) != 0:
        kname = "k_project_lin"                          # try_launch_lin of csrc/project_lin.hip
    if a.ingest == "u8":
        kname = "k_project_u8_max" if Z % 16 == 0 else "k_project_fast<uint8>"
    roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "launches": int(launches), "avg_launch_ms": round(avg_ms, 4), "frames_per_launch": frames_per_launch,
                "algorithmic_bytes_per_frame": alg_bytes_frame}

    # ---- the GEMM alone (one pipeline chunk of code rows, nothing else on the GPU) ------------------------
    #      "alone": the 128x128 kernel the pipeline runs beside the projection, one pipeline chunk; "alone_large_batch": the code
    #      rows of (up to) 65 536 frames through rml_svm_decision's own chunking (k_svm_gemm_ring: 256x256 tiles, chunks sized
    #      for whole rounds of one workgroup per CU), the rate a caller of decision_function on a large batch of rows gets
    if groof is not None:
        for key, nb in (("alone", int(min(frames_per_launch, B))), ("alone_large_batch", int(min(65536, B)))):
            _, q, isum, isq, flags = rml.process_volumes(V[:nb], mode="max", scale=True, codes=True)
            if q.stride(0) % 16 == 0:
                for _ in range(2):
                    svc.decide_codes(q, isum, isq, flags, want_proba=True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    svc.decide_codes(q, isum, isq, flags, want_proba=True)
                e1.record()
                e1.synchronize()
                alone_ms = e0.elapsed_time(e1) / 5
                alone = 2.0 * D * M * nb / (alone_ms * 1e-3) / 1e12
                groof[key] = {"frames": nb, "ms": round(alone_ms, 4), "achieved": round(alone, 1),
                              "frac": round(alone / I8_MFMA_PEAK_TOPS, 4),
                              "kernel": "k_svm_gemm_ring<PT,0> + k_svm_finish" if key == "alone_large_batch" else "k_svm_gemm<I8> + k_svm_finish"}
            del q, isum, isq, flags

    # ---- configs[1]: the projection kernel alone (HBM-roofline check; float32 feature rows written) ------------
    proj_only = None
    if primary:
        proj_only = {}
        for nb in (4096, 16384):
            nb = int(min(nb, B))
            feat_o = torch.empty((nb, D), dtype=torch.float32, device=dev)
            for _ in range(3):
                rml.process_volumes(V[:nb], mode="max", out=feat_o)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                rml.process_volumes(V[:nb], mode="max", out=feat_o)
            e1.record()
            e1.synchronize()
            ms_p = e0.elapsed_time(e1) / 10
            gbs = nb * (frame_bytes + 4 * D) / (ms_p * 1e-3) / 1e9
            proj_only["batch_%d" % nb] = {"ms": round(ms_p, 4), "frames_per_s": round(nb / ms_p * 1e3), "achieved_GBs": round(gbs, 1),
                                          "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_frame": frame_bytes + 4 * D}
            del feat_o

    # ---- parity gate on a slab of the very frames the GPU classified ----------------------
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c as OC
    npar = min(a.parity if primary else min(a.parity, 1024), B)
    vh = V[:npar].cpu().numpy().astype(np.float32)
    threads = len(os.sched_getaffinity(0))
    xz, yz, xy = OC.project_max(vh, threads=threads)
    fh = OC.features(xz, yz, xy, scale=True)
    ref = OC.svm(fh, sv_f64(model), model["dual_coef"], model["intercept"], model["n_support"], model["gamma"],
                 "rbf", model["calib_a"], model["calib_b"], threads=threads)
    parity = {
        "frames": int(npar),
        "label_vote_mismatch": int((out["label_vote"][:npar].cpu().numpy() != ref["label_vote"]).sum()),
        "label_calib_mismatch": int((out["label_calib"][:npar].cpu().numpy() != ref["label_calib"]).sum()),
        "dec_ovo_max_abs_err": float(np.abs(out["dec_ovo"][:npar].cpu().numpy() - ref["dec_ovo"]).max()),
        "dec_ovr_max_abs_err": float(np.abs(out["dec_ovr"][:npar].cpu().numpy() - ref["dec_ovr"]).max()),
        "proba_max_abs_err": float(np.abs(out["proba"][:npar].cpu().numpy() - ref["proba"]).max()),
        "accuracy_vs_synth_class": float((out["label_calib"][:npar].cpu().numpy() ==
                                          np.searchsorted(model["classes"], cls[:npar].cpu().numpy())).mean()),
    }

    if slice_rows:
        nsl = int(min(512 if primary else 256, B, npar))
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_np as ONP
        ijk_ref = np.array([[t.i, t.j, t.k] for v in vh[:nsl] for t in ONP.get_derived_targets(v, X, Y, Z)], dtype=np.int32)
        sxz, syz, sxy = OC.project_slice(vh[:nsl], ijk_ref)
        sref = OC.svm(OC.features(sxz, syz, sxy, scale=True), sv_f64(model), model["dual_coef"], model["intercept"], model["n_support"],
                      model["gamma"], "rbf", model["calib_a"], model["calib_b"], threads=threads)
        for key, r in slice_rows.items():
            o_s = r.pop("_out")
            r["parity"] = {
                "frames": nsl,
                "label_vote_mismatch": int((o_s["label_vote"][:nsl].cpu().numpy() != sref["label_vote"]).sum()),
                "label_calib_mismatch": int((o_s["label_calib"][:nsl].cpu().numpy() != sref["label_calib"]).sum()),
                "dec_ovo_max_abs_err": float(np.abs(o_s["dec_ovo"][:nsl].cpu().numpy() - sref["dec_ovo"]).max()),
                "proba_max_abs_err": float(np.abs(o_s["proba"][:nsl].cpu().numpy() - sref["proba"]).max())}
            if "ijk" in o_s:
                # compared by energy: the reference's argpartition leaves exact ties open
                gi = o_s["ijk"][:nsl].cpu().numpy()
                bad = 0
                for b in range(nsl):
                    prof = ONP.axis_energy_profiles(vh[b])
                    bad += int(any(prof[ax][gi[b, ax]] != prof[ax][ijk_ref[b, ax]] for ax in range(3)))
                r["parity"]["derived_target_mismatch"] = bad
            del o_s

    # ---- CPU baseline (rank 0): the reference's own path with the reference's own libraries -- numpy max ->
    #      scipy.ndimage.zoom(.,1.0) + concatenate (common.process_samples) -> sklearn CalibratedClassifierCV(SVC(rbf)).predict
    #      (SURVEY.md §8d), single-process and on all host cores; the C port of the oracle is kept as a second figure ---
    cpu = None
    if not a.no_cpu:                    # rank 0, every N (north_star: "timed in the same run"); the other ranks wait at the next barrier
        BS = _bench_support()
        gl = out["label_calib"][:npar].cpu().numpy()
        try:
            cpu = BS.reference_libs_baseline(vh, model, gl, threads, budget_s=25.0 if primary else 12.0,
                                             want_all=2048 if primary else 1024)
        except Exception as e:                                  # sklearn internals moved: say so, keep the port
            cpu = {"value": None, "unit": "frames/s", "cores": threads, "kind": "port", "port_of": "the reference's own library calls (NumPy max, scipy.ndimage.zoom, scikit-learn CalibratedClassifierCV(SVC).predict), restated call for call", "error": repr(e)[:200]}
        ncpu = a.cpu_frames
        if ncpu <= 0:
            est = 3.0 * D * M / (1.2e9 * threads)
            ncpu = int(max(threads * 4, min(npar, (6.0 if primary else 3.0) / max(est, 1e-6))))
        ncpu = min(ncpu, npar)
        t1 = time.perf_counter()
        cxz, cyz, cxy = OC.project_max(vh[:ncpu], threads=threads)
        cf = OC.features(cxz, cyz, cxy, scale=True)
        OC.svm(cf, sv_f64(model), model["dual_coef"], model["intercept"], model["n_support"], model["gamma"], "rbf",
               model["calib_a"], model["calib_b"], threads=threads)
        cdt = time.perf_counter() - t1
        cpu["c_port"] = {"value": round(ncpu / cdt, 2), "unit": "frames/s", "cores": threads, "kind": "port",
                       "sample": "%d of the same frames, oracle/oracle.c (max-projection + float64 libsvm loops, OpenMP over frames), "
                                 "%.1f s" % (ncpu, cdt)}

    # ---- one observation per call: the grain the reference calls at (predict.py:98-119), host clock per call -------------------
    single = None
    if rank == 0 and a.ingest != "u8":
        def _lat(fn, n=200):
            for _ in range(10):
                fn()
            torch.cuda.synchronize(dev)
            ts = []
            for _ in range(n):
                t1 = time.perf_counter(); fn(); torch.cuda.synchronize(dev); ts.append(time.perf_counter() - t1)
            return round(float(np.percentile(np.array(ts) * 1e6, 50)), 1)
        v1 = V[:1].contiguous()
        v1u = v1.to(torch.uint8)
        ijk1 = torch.tensor([[X // 2, Y // 2, Z // 3]], dtype=torch.int32, device=dev)
        o1 = svc.decide_volumes(v1, mode="max", scale=True)
        single = {"unit": "us per call (p50 of 200, host clock, B = 1)",
                  "f32_volume_to_labels": _lat(lambda: svc.decide_volumes(v1, mode="max", scale=True)),
                  "u8_volume_to_labels": _lat(lambda: svc.decide_volumes(v1u, mode="max", scale=True)),
                  "slice_at_sdk_target_to_labels": _lat(lambda: svc.decide_volumes(v1, mode="slice", ijk=ijk1, scale=True, validate_ijk=False)),
                  "batch_64_volumes_to_labels": _lat(lambda: svc.decide_volumes(V[:64], mode="max", scale=True), 100),
                  "same_bits_as_in_the_batch": bool(torch.equal(o1["dec_ovo"], out["dec_ovo"][:1]) and torch.equal(o1["label_calib"], out["label_calib"][:1])),
                  "note": "DESIGN.md 3.7: the frame split over the chip + matrix-vector SVM kernels, everything on one stream"}

    import zlib
    all_lab = out["all_labels"] if world > 1 else out["label_calib"]
    labels_crc = int(zlib.crc32(all_lab.cpu().numpy().astype(np.int32).tobytes()))      # of every frame's label, in global frame order
    value = world * B * a.steps / dt
    res = {
        "value": round(value, 1), "ms_per_step": round(dt / a.steps * 1e3, 3), "labels_crc32": labels_crc,
        "config": {"workload": "configs[2]: max-projection + RBF-SVM decision_function, 3-class, %d SVs, "
                               "batch %d frames/GPU of %dx%dx%d f32 resident in HBM" % (M, B, X, Y, Z),
                   "grid": [X, Y, Z], "frames_per_gpu": B, "global_frames": world * B, "n_sv": M, "D": D,
                   "gamma": a.gamma, "parallelism": "frames sharded x%d, labels all-gathered (RCCL)" % world
                   if world > 1 else "single GPU"},
        "hbm_frac_end_to_end": round(value / world * (frame_bytes + 16) / 1e9 / HBM_PEAK_GBS, 4),
        "roofline": roofline, "gemm_roofline": groof, "projection_only": proj_only, "cpu_baseline": cpu, "parity": parity,
        "uint8_ingest": u8, "slice_rows": slice_rows, "single_observation": single,
        "model": {"fit_s": round(fit_s, 1), "val_acc": model["val_acc"], "kernel_nondegenerate_frac": model["kfrac"]},
    }
    del V, Vall, out, svc
    torch.cuda.empty_cache()
    return res


def irregular_rows(rml, torch, feat, grid, seed):
    """Feature rows off the code grid the way the reference makes them (train.py:84-185, driven at train.py:496-517): every
    projection of a sample rotated by U(-15, 15) degrees (first third of the rows), zoomed by one U(0.7, 1.3) factor per sample
    (second third) or hit by sparse Gaussian noise of sd 0.2 (last third) -- order-3 spline resampling and the [0, 1] clamp
    through rml_augment on the device.  feat: (N, D) float32 CUDA rows in [0, 1] (scaled max-projections)."""
    X, Y, Z = grid
    N = feat.shape[0]
    rng = np.random.default_rng(seed)
    shapes = ((X, Z), (Y, Z), (X, Y))
    offs = np.cumsum([0] + [h * w for h, w in shapes])
    third = (N // 3, 2 * (N // 3))
    out = torch.empty_like(feat)
    zf = rng.uniform(0.7, 1.3, N)
    for pi, (h, w) in enumerate(shapes):
        planes = feat[:, offs[pi]:offs[pi + 1]].reshape(N, h, w).contiguous()
        dst = out[:, offs[pi]:offs[pi + 1]]
        ang = rng.uniform(-15.0, 15.0, third[0])
        par = np.stack([rml.rotation_params(v, (h, w)) for v in ang]) if third[0] else np.zeros((0, 6))
        if third[0]:
            dst[:third[0]] = rml.augment_planes(planes[:third[0]], "rotate", par).reshape(third[0], -1)
        if third[1] > third[0]:
            dst[third[0]:third[1]] = rml.augment_planes(planes[third[0]:third[1]], "zoom", zf[third[0]:third[1]]).reshape(third[1] - third[0], -1)
        if N > third[1]:
            dst[third[1]:] = rml.augment_planes(planes[third[1]:], "noise", rng.normal(0.0, 0.2, N - third[1])).reshape(N - third[1], -1)
    return out


def run_general(a, env, grid, frames):
    """Rows that are NOT on the integer code grid -- what train.py:496-517 augmentation, a non-unit proj_zoom
    (predict.py:109-116) and the reference's shipped generated_data pickles hold: the projections of the synthetic frames are
    rotated / zoomed / noised by rml_augment (spline resampling: arbitrary float32 values), the support vectors likewise, and
    the step is clf.predict_proba on those rows (rml_svm_decision).  RML_PATH_AUTO takes the multi-digit int8 kernel
    (k_svm_gemm_ring<PT, 1>: four balanced int8 digits per value, ten exact digit-plane products); the float64-MFMA kernel
    (v_mfma_f64_16x16x4_f64) is timed beside it.  Also the fused front door on off-grid VOLUMES (projection -> float rows ->
    digit planes -> GEMM).  Parity vs the float64 C oracle on the very rows."""
    import torch
    import torch.distributed as dist
    rml, _lib, dev, rank, world = env["rml"], env["lib_mod"], env["dev"], env["rank"], env["world"]
    X, Y, Z = grid
    D = rml.feature_len(X, Y, Z)
    obj = [None]
    if rank == 0:
        obj[0] = fit_model(rml, torch, X, Y, Z, a.train, a.gamma, a.seed, dev)
    if world > 1:
        dist.broadcast_object_list(obj, src=0)
    model = obj[0]
    sv_rows = torch.from_numpy(sv_f64(model).astype(np.float32)).to(dev)
    sv = irregular_rows(rml, torch, sv_rows, grid, a.seed + 11).cpu().numpy().astype(np.float64)
    del sv_rows
    svc = rml.GpuSVC(sv, model["dual_coef"], model["intercept"], model["n_support"], model["gamma"], model["classes"],
                     calib_a=model["calib_a"], calib_b=model["calib_b"])
    assert not svc.exact
    M = int(sv.shape[0])
    B = int(frames)
    V, _ = rml.synth_volumes(B, X, Y, Z, seed=a.seed + 3, frame0=rank * B, device=dev)
    rows = irregular_rows(rml, torch, rml.process_volumes(V, mode="max", scale=True), grid, a.seed + 12 + rank)
    off_grid = float((torch.round(rows[:256] * 255.0) / 255.0 != rows[:256]).float().mean())
    lib = _lib.load()
    ctx = _lib.context(dev)
    from radar_ml_amd import dist as rdist

    def timed(fn, steps):
        for _ in range(max(1, a.warmup)):
            o = fn()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            o = fn()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        tm = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        return o, float(tm.item())

    def step_rows(path):
        def f():
            ovo, ovr, vote, proba, lab = svc._decide(rows, want_proba=True, path=path)
            if world > 1:
                rdist.gather_labels(lab)
            return {"dec_ovo": ovo, "label_vote": vote, "proba": proba, "label_calib": lab}
        return f

    out, dt = timed(step_rows("auto"), a.steps)
    out64, dt64 = timed(step_rows("f64"), max(1, a.steps // 4))
    # the fused front door on volumes whose values left the grid (uniform 1 - 2^-10 scale + noise on the returns)
    V.mul_(0.9990234375)
    V.add_(torch.randn_like(V) * 0.05 * (V > 0))
    outv, dtv = timed(lambda: svc.decide_volumes(V, mode="max", scale=True, want_proba=True), max(1, a.steps // 2))
    if rank != 0:
        return None
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c as OC
    npar = min(1024, B)
    threads = len(os.sched_getaffinity(0))
    # rows from all three kinds (rotated / zoomed / noised)
    pick = np.concatenate([np.arange(0, npar // 3), B // 3 + np.arange(0, npar // 3), B - (npar - 2 * (npar // 3)) + np.arange(0, npar - 2 * (npar // 3))])
    fh = rows[torch.as_tensor(pick, device=dev)].cpu().numpy()
    ref = OC.svm(fh, sv, model["dual_coef"], model["intercept"], model["n_support"], model["gamma"], "rbf",
                 model["calib_a"], model["calib_b"], threads=threads)
    vh = V[:256].cpu().numpy()
    xz, yz, xy = OC.project_max(vh, threads=threads)
    refv = OC.svm(OC.features(xz, yz, xy, scale=True), sv, model["dual_coef"], model["intercept"], model["n_support"], model["gamma"], "rbf",
                  model["calib_a"], model["calib_b"], threads=threads)

    def par(o, r, idx):
        g = {k: o[k][torch.as_tensor(idx, device=dev)].cpu().numpy() for k in ("dec_ovo", "label_vote", "proba", "label_calib")}
        return {"frames": int(len(idx)), "label_vote_mismatch": int((g["label_vote"] != r["label_vote"]).sum()),
                "label_calib_mismatch": int((g["label_calib"] != r["label_calib"]).sum()),
                "dec_ovo_max_abs_err": float(np.abs(g["dec_ovo"] - r["dec_ovo"]).max()),
                "proba_max_abs_err": float(np.abs(g["proba"] - r["proba"]).max())}

    # throughput of the augmentation kernels themselves (train.py:84-185 on the GPU): device-resident xz planes of the batch,
    # one rml_augment launch per kind; bytes = one float32 read + one write per pixel
    aug = None
    if rank == 0:
        planes = rows[:, :X * Z].reshape(B, X, Z).contiguous()
        rng = np.random.default_rng(a.seed + 5)
        pars = {"rotate": np.stack([rml.rotation_params(v, (X, Z)) for v in rng.uniform(-15.0, 15.0, B)]),
                "zoom": rng.uniform(0.7, 1.3, B), "noise": rng.normal(0.0, 0.2, B)}
        aug = {"workload": "%d device-resident %dx%d float32 planes per launch (csrc/augment.hip: order-3 spline rotate / clipped zoom with "
                           "float64 arithmetic like SciPy, sparse noise), parameters uploaded per call" % (B, X, Z)}
        for kind in ("rotate", "zoom", "noise"):
            rml.augment_planes(planes, kind, pars[kind])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                rml.augment_planes(planes, kind, pars[kind])
            torch.cuda.synchronize()
            dta = (time.perf_counter() - t0) / 3
            aug[kind] = {"planes_per_s": round(B / dta), "ms": round(dta * 1e3, 3), "GBs": round(B * X * Z * 8 / dta / 1e9, 1)}
        del planes
    ops = 2.0 * D * M
    value = world * B * a.steps / dt
    v64 = world * B * max(1, a.steps // 4) / dt64
    vv = world * B * max(1, a.steps // 2) / dtv
    return {
        "value": round(value, 1), "unit": "frames/s", "ms_per_step": round(dt / a.steps * 1e3, 3),
        "dtype": "f32 rows -> 4 balanced int8 digits of a 32-bit fixed-point value, i8 MFMA exact int32, f64 epilogue",
        "workload": "%d rows/GPU of D = %d (max-projections of %dx%dx%d frames, every projection rotated / spline-zoomed / noised by "
                    "rml_augment: %.0f %% of the values off the code grid), %d SVs augmented the same way; clf.predict_proba on the rows: "
                    "RML_PATH_AUTO -> multi-digit int8 kernel" % (B, D, X, Y, Z, 100.0 * off_grid, M),
        "roofline": {"bound": "mfma", "kernel": "k_svm_gemm_ring<PT,1> (+ k_prepare_rows, k_digit_rows, k_svm_finish)",
                     "achieved": round(value / world * 10.0 * ops / 1e12, 1), "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s",
                     "frac": round(value / world * 10.0 * ops / 1e12 / I8_MFMA_PEAK_TOPS, 4),
                     "note": "ten digit-plane products of 2*D*M int8 ops per row, row preparation inside the timed step"},
        "parity": par(out, ref, pick),
        "float64_mfma_path": {"value": round(v64, 1), "unit": "frames/s", "kernel": "k_svm_gemm<F64> (v_mfma_f64_16x16x4_f64)",
                              "achieved_TFLOPs": round(v64 / world * ops / 1e12, 2), "frac_of_78.6": round(v64 / world * ops / 1e12 / 78.6, 4),
                              "parity": par(out64, ref, pick),
                              "max_abs_dec_diff_vs_digits": float((out64["dec_ovo"] - out["dec_ovo"]).abs().max())},
        "speedup_vs_float64_mfma": round(value / v64, 2),
        "from_volumes": {"value": round(vv, 1), "unit": "frames/s",
                         "workload": "the fused front door on the %d volumes scaled by 1 - 2^-10 with N(0, 0.05) noise on the returns "
                                     "(projection -> float rows -> digit planes -> GEMM)" % B,
                         "parity": par(outv, refv, np.arange(256))},
        "augmentation": aug,
    }


def _top2_margin(p):
    srt = np.sort(p, axis=1)
    return srt[:, -1] - srt[:, -2]


def dnn_bench_model(rml, dev, seed, train_steps, grid=(22, 31, 176)):
    """The CNN row's model: the reference's architecture (dnn.py:45-91) with trained weights -- plumbing: float32 Adam steps with
    plain PyTorch layers, dnn.py:89-90 optimizer, on synthetic 3-class frames through the reference's preprocessing; random-init
    outputs sit at ~1/3 each and say nothing about labels.  Every rank trains the same model from the same seed."""
    import importlib
    import torch
    dnn = importlib.import_module("radar_ml_amd.dnn")
    nn_common = importlib.import_module("radar_ml_amd.nn_common")
    X, Y, Z = grid
    torch.manual_seed(seed)
    model = dnn.define_classifier(device=dev)
    tv, tcls = rml.synth_volumes(1024, X, Y, Z, seed=seed + 5, frame0=1 << 41, device=dev)
    tfeat = rml.process_volumes(tv, mode="max", scale=False)
    txs = [t.reshape(-1, 1, 80, 80) for t in nn_common.preprocess_features(tfeat, (X, Y, Z), (80, 80), out_dtype="float32")]
    ty = tcls.to(dev).long()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.5, 0.999), eps=1e-7)
    gtr = torch.Generator(device=dev).manual_seed(seed)
    model.train()
    # deterministic convolution algorithms while training: the same model in every run (MIOpen's default weight-gradient kernels
    # accumulate with atomics: mean margin 0.45 ... 0.63 and 390 ... 1 280 guard rows from run to run of the same seed, session r5m)
    det_was = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        for _ in range(train_steps):
            idx = torch.randint(0, 1024, (64,), device=dev, generator=gtr)
            opt.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(model.logits(*[t[idx] for t in txs]).float(), ty[idx]).backward()
            opt.step()
    finally:
        torch.backends.cudnn.deterministic = det_was
    model.eval()
    return model


def run_dnn(a, env):
    """BASELINE configs[3]: multi-view CNN inference at the Walabot arena grid -- projection into uint8 code rows (csrc/project_lin.hip;
    a device-predicated float pass for frames off the code grid) -> [-1,1] scaling + bicubic resize to 80x80 of the three projections
    in one launch (csrc/preprocess.hip: Pillow's windows and weights in float32, bf16 out) -> fused conv trunk (csrc/dnn.hip) ->
    fused dense tail (csrc/dense.hip), bf16 operands, random-init weights of the reference's architecture (dnn.py:45-91).  The
    parity leg compares with the float64 NumPy restatement of the reference's chain (Pillow-exact resize included).
    Frames are sharded over the ranks; the predicted labels are all-gathered (RCCL) when N > 1.  Returns the result dict
    on rank 0."""
    import importlib
    import torch
    import torch.distributed as dist
    rml, dev, rank, world = env["rml"], env["dev"], env["rank"], env["world"]
    dnn = importlib.import_module("radar_ml_amd.dnn")
    X, Y, Z = 22, 31, 176
    B = a.dnn_frames
    model = dnn_bench_model(rml, dev, a.seed, a.dnn_train_steps, (X, Y, Z))
    V, Vcls = rml.synth_volumes(B, X, Y, Z, seed=a.seed + 7, frame0=rank * B, device=dev)
    from radar_ml_amd import dist as rdist

    trunk_ev = {"f32": [], "u8": []}

    def step(vol, evs=None):
        p = model.predict_volumes(vol, trunk_events=evs)
        if world > 1:
            rdist.gather_labels(p.argmax(dim=1).to(torch.int32))          # RCCL all-gather of the predictions, 4 B/frame
        return p

    res = {}
    guard = {}
    for tag, vol in (("f32", V), ("u8", V.to(torch.uint8)), ("f32_noguard", V)):
        if tag == "f32_noguard":          # the same steps without the margin guard: what the float64 labels cost

            def step(vol, evs=None):
                p = model.predict_volumes(vol, label_guard=None)
                if world > 1:
                    rdist.gather_labels(p.argmax(dim=1).to(torch.int32))
                return p
        for _ in range(max(1, a.warmup)):
            p = step(vol)
        guard[tag] = dict(model.last_guard)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            p = step(vol, trunk_ev.get(tag))             # an event pair around every trunk launch INSIDE the timed steps
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        res[tag] = (float(dt.item()), p)
    # the same frames through the round-1..3 front of the chain (float rows, Pillow-bit-identical float64 resize, one launch per
    # projection): what the float32 preprocessing kernel is allowed to change is the bf16 rounding of ~3e-4 of the pixels
    for _ in range(2):
        pe = model.predict_volumes(V, exact_resize=True, label_guard=None)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pe = model.predict_volumes(V, exact_resize=True, label_guard=None)
    torch.cuda.synchronize(dev)
    exact_dt = time.perf_counter() - t0
    exact_dp = float((pe - res["f32_noguard"][1]).abs().max())
    # the guard's worst case: untrained weights of the same architecture -- outputs ~1/3 each, nearly every row inside the gap, so
    # nearly every row is scored twice (bf16 chain, then the float32-class trunk on exact inputs)
    torch.manual_seed(a.seed + 3)
    rmodel = dnn.define_classifier(device=dev).eval()
    for _ in range(2):
        rp = rmodel.predict_volumes(V)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        rp = rmodel.predict_volumes(V)
    torch.cuda.synchronize(dev)
    rdt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(rdt, op=dist.ReduceOp.MAX)
    rguard = dict(rmodel.last_guard)
    if rank != 0:
        return None
    # the guard's bound is empirical (a row OUTSIDE the covered gap whose bf16 error exceeds half its margin would keep a wrong
    # label): check it on every frame of the timed batch -- all of them once more through the float32-class trunk on exact inputs
    # (outside the timed region), labels compared wherever that chain itself is not at a tie
    with torch.no_grad():
        px3 = model.rescore_exact(V, precision="x3")
        pg, pn = res["f32"][1].float(), res["f32_noguard"][1].float()
        g3 = model._gaps(px3)
        sure = g3 >= dnn.LABEL_GUARD_X3
        outside = model._gaps(pn) >= float(guard["f32"].get("gap") or dnn.LABEL_GUARD)
        whole = {"rows": int(B), "label_mismatch_vs_float32_class_chain": int((px3.argmax(1) != pg.argmax(1))[sure].sum()),
                 "rows_at_a_float32_tie": int((~sure).sum()),
                 "label_mismatch_without_guard": int((px3.argmax(1) != pn.argmax(1))[sure].sum()),
                 "max_bf16_error_outside_the_gap": float((pn - px3.float()).abs().max(1).values[outside].max()) if bool(outside.any()) else 0.0,
                 "max_bf16_error": float((pn - px3.float()).abs().max()),
                 "half_gap": 0.5 * float(guard["f32"].get("gap") or dnn.LABEL_GUARD)}
        del px3
    # parity on a few frames: NumPy restatement of the whole chain (oracle projections, Pillow restatement, Keras layers)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_np as O
    npar = int(min(max(8, a.dnn_parity), B))
    vh = V[:npar].cpu().numpy()
    planes = [[], [], []]
    for v in vh:
        for i, pr in enumerate(O.project_max(v)):
            planes[i].append(O.pil_resize_bicubic(O.scale_unit_range(pr), (80, 80)))
    convs, dense = model.keras_weights()
    want = O.dnn_forward(np.stack(planes[0]), np.stack(planes[1]), np.stack(planes[2]), convs, dense)
    got = res["f32"][1][:npar].float().cpu().numpy()
    nrp = int(min(256, npar))
    rwant = O.dnn_forward(np.stack(planes[0][:nrp]), np.stack(planes[1][:nrp]), np.stack(planes[2][:nrp]), *rmodel.keras_weights())
    rgot = rp[:nrp].float().cpu().numpy()
    # roofline of the dominant kernel of this row, k_dnn_trunk_rf (bf16 MFMA): the event pairs recorded around EVERY trunk launch of
    # the timed steps above, on the stream it runs on (round 3 timed 13 launches after seconds of host-only oracle work: the chip
    # had clocked down, and the line disagreed with the rocprofv3 summary by 30 %); full batches only
    full = [(e0.elapsed_time(e1), nfr) for e0, e1, nfr in trunk_ev["f32"]]
    nb = max(nfr for _, nfr in full)
    ms_l = [ms for ms, nfr in full if nfr == nb]
    trunk_ms = float(np.mean(ms_l))
    conv_flop = 3 * 2.0 * (40 * 40 * 64 * 9 + 20 * 20 * 32 * 576)            # per frame: 3 branches x (conv1 + conv2), dnn.py:45-52
    trunk_tf = conv_flop * nb / (trunk_ms * 1e-3) / 1e12
    BF16_PEAK = 2500.0                                                         # MI355X_MICROARCH.md: dense bf16 MFMA
    out = {"metric": "radar frames/s (3D-proj->resize->CNN forward)", "unit": "frames/s", "dtype": "bf16 operands, f32 accumulate",
           "value": round(world * B * a.steps / res["f32"][0], 1), "ms_per_step": round(res["f32"][0] / a.steps * 1e3, 3),
           "value_uint8_volumes": round(world * B * a.steps / res["u8"][0], 1),
           "pillow_exact_resize_chain": {"value_this_rank": round(B * a.steps / exact_dt, 1), "proba_max_abs_diff_vs_fused_preprocessing": exact_dp},
           "uint8_identical_labels": bool(torch.equal(res["u8"][1].argmax(1), res["f32"][1].argmax(1))),
           "config": {"workload": "configs[3]: %d frames/GPU of %dx%dx%d -> 3 x 80x80 -> multi-view CNN (2.52 M parameters, "
                                  "54.7 MFLOP/frame), weights trained for %d Adam steps on synthetic frames (accuracy on the "
                                  "timed frames %.3f)" % (B, X, Y, Z, a.dnn_train_steps, float((res["f32"][1].argmax(1).cpu() == Vcls.cpu()).float().mean())),
                      "frames_per_gpu": B},
           "roofline": {"bound": "mfma", "kernel": "k_dnn_trunk_rf", "achieved": round(trunk_tf, 1), "peak": BF16_PEAK, "unit": "TFLOP/s",
                        "frac": round(trunk_tf / BF16_PEAK, 4), "avg_launch_ms": round(trunk_ms, 4), "frames_per_launch": nb,
                        "launches": len(ms_l), "min_launch_ms": round(float(np.min(ms_l)), 4), "max_launch_ms": round(float(np.max(ms_l)), 4),
                        "algorithmic_flop_per_frame": conv_flop, "traffic": None},
           # the margin guard (Classifier.predict_volumes: rows whose top-2 gap is below dnn.LABEL_GUARD are scored again in float64)
           "margin_guard": {"gap_initial": dnn.LABEL_GUARD, "gap_used": guard["f32"].get("gap"), "observed_bf16_error": guard["f32"].get("observed_error"),
                            "stages": "bf16 chain -> x3 (bf16 operand pairs, csrc/dnn_x3.hip) below the gap -> x6 (triples) below %g -> float64 below %g"
                                      % (dnn.LABEL_GUARD_X3, dnn.LABEL_GUARD_X6),
                            "rows": guard["f32"]["rows"], "rescored": guard["f32"]["rescored"], "rescored_x6": guard["f32"].get("rescored_x6"),
                            "rescored_float64": guard["f32"]["rescored_float64"], "observed_x3_error": guard["f32"].get("observed_error_x3"),
                            "rounds": guard["f32"].get("rounds"), "covered": guard["f32"].get("covered"),
                            "value_without_guard": round(world * B * a.steps / res["f32_noguard"][0], 1),
                            "cost_frac": round(res["f32"][0] / res["f32_noguard"][0] - 1.0, 4),
                            "label_mismatch_without_guard": int((res["f32_noguard"][1][:npar].float().cpu().numpy().argmax(1) != want.argmax(1)).sum()),
                            "whole_batch_check": whole},
           "random_init": {"value": round(world * B * a.steps / float(rdt.item()), 1), "unit": "frames/s",
                           "note": "untrained weights (outputs ~1/3 each): the guard's worst case, nearly every row scored twice",
                           "rescored": rguard["rescored"], "rescored_x6": rguard.get("rescored_x6"), "rescored_float64": rguard["rescored_float64"],
                           "rows": rguard["rows"], "observed_bf16_error": rguard.get("observed_error"),
                           "parity": {"frames": nrp, "label_mismatch": int((rgot.argmax(1) != rwant.argmax(1)).sum()),
                                      "proba_max_abs_err_vs_float64_oracle": float(np.abs(rgot - rwant).max())}},
           "parity": {"frames": npar, "proba_max_abs_err_vs_float64_oracle": float(np.abs(got - want).max()),
                      "label_mismatch": int((got.argmax(1) != want.argmax(1)).sum()),
                      "rows_inside_the_guard_gap": int((_top2_margin(want) <= dnn.LABEL_GUARD).sum()),
                      "mean_top2_margin": float(_top2_margin(want).mean()),
                      "note": "parity unpinned by the reference (no weights, no TensorFlow here): bf16 GPU chain + float64 margin guard "
                              "vs the float64 NumPy restatement of the same chain on the same trained weights; labels on EVERY row"}}
    return out


def run_sgan(a, env, n=256, hw=128, steps=None):
    """BASELINE configs[4]: the SGAN discriminator/classifier train step (c_model + d_model(real, class-weighted) + d_model(fake)
    updates, sgan.py:525-532) on
    128x128 projections, fp16 autocast with loss scaling, MIOpen convolutions for layers 2-3 + csrc/bnact.hip for the rest.
    Data parallel when N > 1: every rank trains on its own batch of ``n`` samples (weak scaling) and the gradients are
    all-reduced once per update on one flat 7.4 MB bucket (RCCL over xGMI) between the HIP-graph replay of forward + backward
    and the optimizer step.  Runs on the CPU too (gloo, no autocast, no graph): that is what tests/test_dist_cpu.py drives
    with two ranks.  Returns the row on rank 0."""
    import importlib
    import torch
    import torch.distributed as dist
    dev, rank, world = env["dev"], env["rank"], env["world"]
    on_gpu = torch.device(dev).type == "cuda"
    sgan = importlib.import_module("radar_ml_amd.sgan")
    torch.manual_seed(a.seed)                                                   # the same initial weights on every rank
    d = sgan.define_discriminator((hw, hw, 1), (hw, hw, 1), (hw, hw, 1), device=dev)
    tr = sgan.DiscriminatorTrainer(d, amp_dtype="float16" if on_gpu else None, use_graph=on_gpu, tune_convolutions=on_gpu)    # ddp: on when world > 1
    g = torch.Generator(device=dev).manual_seed(a.seed + 17 * rank)             # every rank its own shard of the global batch
    x = [torch.rand((n, hw, hw), device=dev, generator=g) * 2 - 1 for _ in range(3)]
    y = torch.randint(0, 3, (n,), device=dev, generator=g)
    # the d updates of sgan.py:529-532: real batch with smoothed positive labels in [0.7, 1.2) (sgan.py:396-398) and the data
    # set's class weights (class_weight -> per-sample weights the way Keras does it), fake batch (here: another batch of
    # projections -- the generator is out of scope) with smoothed negative labels in [0, 0.3) (sgan.py:401-403)
    yr = 0.7 + 0.5 * torch.rand((n, 1), device=dev, generator=g)
    sw = torch.from_numpy(sgan.class_weight_to_sample_weight(yr.cpu().numpy(), {0: 1.0, 1: 1.37, 2: 2.05})).to(dev)
    xf = [torch.rand((n, hw, hw), device=dev, generator=g) * 2 - 1 for _ in range(3)]
    yf = 0.3 * torch.rand((n, 1), device=dev, generator=g)
    for _ in range(6 if on_gpu else 1):
        tr.train_on_batch_c(x, y)
        tr.train_on_batch_d(x, yr, sample_weight=sw)
        tr.train_on_batch_d(xf, yf)

    def fence():
        if on_gpu:
            torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize(dev)

    steps = steps if steps is not None else max(30, a.steps)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        lc, acc = tr.train_on_batch_c(x, y, sync=False)
        ld = tr.train_on_batch_d(x, yr, sample_weight=sw, sync=False)
        lf = tr.train_on_batch_d(xf, yf, sync=False)
    fence()
    dt = torch.tensor([(time.perf_counter() - t0) / steps], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    # replicas must still be identical after the timed updates (sum of |parameters| agrees to the last bit)
    chk = torch.stack([p.detach().double().abs().sum() for p in d.parameters()]).sum().reshape(1)
    same = True
    if world > 1:
        both = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(both, chk)
        same = all(bool(torch.equal(b, both[0])) for b in both)
    if rank != 0:
        return None
    # algorithmic FLOPs of one update on one sample (sgan.py:132-217): per branch three 3x3 stride-2 convolutions (1 -> 128 -> 64 ->
    # 32 channels, hw -> hw/8), then the dense layers; a training update = forward + data gradient + weight gradient of every
    # layer, the first convolution having no data gradient (its input is the data).  Batch norm / activations / optimizer: not counted.
    fwd, first = 0.0, 0.0
    for (h_, w_, c_) in d.shapes:
        ch = c_
        for li, co in enumerate((128, 64, 32)):
            h_, w_ = -(-h_ // 2), -(-w_ // 2)
            f = 2.0 * h_ * w_ * co * ch * 9
            fwd += f
            first += f if li == 0 else 0.0
            ch = co
    fwd += 2.0 * (d.flat_features * 64 + 64 * 64 + 64 * d.n_classes)
    flop_update = 3.0 * fwd - first
    F16_PEAK = 2500.0                                   # MI355X_MICROARCH.md: dense fp16 MFMA (= bf16)
    tf = 3 * n * flop_update / dt / 1e12
    return {"metric": "sgan discriminator train step (c + d_real[class_weight] + d_fake updates, sgan.py:525-532)",
            "roofline": {"bound": "mfma", "kernel": "MIOpen igemm fwd / bwd / wrw (layers 2-3) + csrc/bnact.hip first layer", "achieved": round(tf, 1),
                         "peak": F16_PEAK, "unit": "TFLOP/s", "frac": round(tf / F16_PEAK, 4),
                         "algorithmic_flop_per_sample_update": flop_update,
                         "note": "whole three-update step over its wall time (launch-bound small convolutions: 128x128 planes, batch 256); "
                                 "the convolutions' own share is in profiles/r05_stats_sgan.txt"},
            "value": round(world * 3 * n / dt, 1), "unit": "samples/s", "updates_per_step": 3,
            "ms_per_step": round(dt * 1e3, 2), "batch_per_gpu": n, "global_batch": world * n, "n_gpus": world,
            "parallelism": "data parallel x%d: one flat-bucket gradient all-reduce (%d parameters) per update"
                           % (world, sum(p.numel() for p in d.parameters())) if world > 1 else "single GPU",
            "replicas_identical": same, "hip_graph": bool(tr.use_graph),
            "dtype": "fp16 autocast (MIOpen convolutions, csrc/bnact.hip batch-norm/activation/pad), fp32 master weights"
                     if on_gpu else "float32 (CPU)",
            "c_loss": round(float(lc), 4), "d_loss": round(float(ld), 4), "d_fake_loss": round(float(lf), 4)}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves (one process per GPU,
    torch.distributed.run on 127.0.0.1, the contract's own command line) and return their exit status.  A request the box
    cannot serve -- fewer visible devices than ranks -- is refused loudly; RML_BENCH_ONE_DEVICE=1 is the declared dry run
    of the N > 1 control flow on one device (gloo)."""
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < a.gpus and not os.environ.get("RML_BENCH_ONE_DEVICE"):
        sys.stderr.write("bench.py: --gpus %d requested but %d device(s) visible; refusing to print an n_gpus=%d line "
                         "(RML_BENCH_ONE_DEVICE=1 runs the ranks on one device over gloo as a control-flow dry run)\n"
                         % (a.gpus, have, a.gpus))
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus is None:
        a.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        # the launcher decides how many ranks exist; a line that claims another n_gpus than asked for is never printed
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or drop the launcher: "
                             "`python bench.py --gpus N` starts its own ranks)\n" % (a.gpus, world, a.gpus))
        sys.exit(2)
    one_device = bool(os.environ.get("RML_BENCH_ONE_DEVICE"))
    if world > 1 and not one_device and torch.cuda.device_count() < world:
        if rank == 0:
            sys.stderr.write("bench.py: %d ranks but %d device(s) visible\n" % (world, torch.cuda.device_count()))
        sys.exit(2)
    if one_device:      # dry run of the N > 1 control flow on a 1-GPU box: every rank on cuda:0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_device:
            dist.init_process_group("gloo")         # RCCL refuses two ranks on one device; gloo stages CUDA tensors through the host
        else:
            dist.init_process_group("nccl", device_id=dev)
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    import radar_ml_amd as rml
    from radar_ml_amd import _lib
    env = {"rml": rml, "lib_mod": _lib, "dev": dev, "rank": rank, "world": world}

    grid = tuple(int(t) for t in a.grid.lower().split("x"))
    res = run_workload(a, env, grid, a.frames, primary=True)
    # north_star: "Throughput ... at the Walabot Arena grid shape is reported": the reference's own grid
    # (22, 31, 176) (common.py:25-27, predict.log:13) as a second workload in the same run, same metric.
    wal = None
    if not a.no_walabot and grid != (22, 31, 176):
        wal = run_workload(a, env, (22, 31, 176), a.walabot_frames, primary=False)

    # roofline.traffic: HBM bytes per projection launch from THIS run's counters (two rocprofv3 PMC passes over a child
    # process that launches the same kernels on the same grids and frames per launch); null when that is not possible
    if rank == 0 and world == 1 and not a.no_pmc:
        cfgs = []
        for tag, r in (("primary", res), ("walabot", wal)):
            if r is None:
                continue
            g = r["config"]["grid"]
            cfgs.append({"tag": tag + "_f32", "grid": g, "frames": int(r["roofline"]["frames_per_launch"]), "u8": False})
            if r.get("uint8_ingest"):
                cfgs.append({"tag": tag + "_u8", "grid": g, "frames": int(r["uint8_ingest"]["roofline"]["frames_per_launch"]), "u8": True})
            for key, mode in (("derive_slice_svm", "derive_slice"), ("slice_mode", "slice")):
                if r.get("slice_rows"):
                    cfgs.append({"tag": tag + "_" + key, "grid": g, "frames": int(r["slice_rows"][key]["roofline"]["frames_per_launch"]),
                                 "u8": False, "mode": mode})
        tr = None
        try:
            torch.cuda.empty_cache()
            tr = _bench_support().measure_traffic(cfgs)
        except Exception:
            tr = None
        if tr:
            for tag, r in (("primary", res), ("walabot", wal)):
                if r is None:
                    continue
                t = tr.get(tag + "_f32")
                if t:
                    r["roofline"]["traffic"] = t["hbm_bytes"]
                    r["roofline"]["traffic_detail"] = {k: t[k] for k in ("fetch_bytes", "write_bytes", "kernel", "source")}
                t = tr.get(tag + "_u8")
                if t and r.get("uint8_ingest"):
                    r["uint8_ingest"]["roofline"]["traffic"] = t["hbm_bytes"]
                    r["uint8_ingest"]["roofline"]["traffic_detail"] = {k: t[k] for k in ("fetch_bytes", "write_bytes", "kernel", "source")}
                for key in ("derive_slice_svm", "slice_mode"):
                    t = tr.get(tag + "_" + key)
                    if t and r.get("slice_rows"):
                        rr = r["slice_rows"][key]["roofline"]
                        rr["traffic"] = t["hbm_bytes"]
                        rr["traffic_over_algorithmic"] = round(t["hbm_bytes"] / (rr["algorithmic_bytes_per_frame"] * rr["frames_per_launch"]), 3)
                        rr["traffic_detail"] = {k: t[k] for k in ("fetch_bytes", "write_bytes", "kernel")}

    if rank == 0 and world > 1:
        for r in (res, wal):
            if r is not None:
                r["roofline"]["traffic_note"] = "not measured at N > 1: the PMC passes run a single-process child; see the N = 1 line"

    gen_row = None
    if not a.no_general:
        gen_row = {}
        g1 = run_general(a, env, grid, a.general_frames)
        if not a.no_walabot and grid != (22, 31, 176):
            g2 = run_general(a, env, (22, 31, 176), a.general_frames * 2)
        else:
            g2 = None
        if rank == 0:
            gen_row = dict(g1)
            if g2 is not None:
                gen_row["walabot_grid"] = g2

    dnn_row = None
    if not a.no_dnn:
        dnn_row = run_dnn(a, env)

    sgan_row = None
    if not a.no_sgan:
        try:
            sgan_row = run_sgan(a, env)
        except Exception as e:          # a library-side failure must not cost the headline line
            sgan_row = {"error": repr(e)[:200]}

    if rank == 0:
        # second denominator of SURVEY.md §8d: what a pure streaming READ reaches on THIS box (rml_probe_stream: a persistent kernel of
        # non-temporal 16-byte loads folded by a max, the projection's own access pattern without its epilogue).  Round 4 used a
        # torch copy_ here, which is slower than the projection it was meant to bound.
        try:
            src = torch.zeros(1 << 30, dtype=torch.uint8, device=dev)
            gbs = ctypes.c_double()
            env["lib_mod"].check(env["lib_mod"].load().rml_probe_stream(env["lib_mod"].context(dev), env["lib_mod"].ptr(src), src.numel(), 20,
                                                                          ctypes.byref(gbs), env["lib_mod"].stream_ptr(dev)), "rml_probe_stream")
            del src
            stream_gbs = float(gbs.value)
            rows = [("primary", res)] + ([("walabot", wal)] if wal is not None else [])
            for _, r in rows:
                for rf_ in [r["roofline"]] + ([r["uint8_ingest"]["roofline"]] if r.get("uint8_ingest") else []) + \
                        [sr["roofline"] for sr in (r.get("slice_rows") or {}).values()]:
                    rf_["measured_stream_GBs"] = round(stream_gbs, 1)
                    rf_["frac_of_measured_stream"] = round(rf_["achieved"] / stream_gbs, 4)
        except Exception as e:
            res["roofline"]["measured_stream_error"] = repr(e)[:160]
        # ---- the output.  The verbose material ("doc") goes on an EARLIER line; the contract keys, the compact roofline /
        #      cpu_baseline objects and a one-screen summary of every row are the LAST line ----
        doc = {"config_detail": res["config"], "roofline": res["roofline"], "gemm_roofline": res["gemm_roofline"],
               "cpu_baseline": res["cpu_baseline"], "parity": res["parity"], "projection_only_configs1": res["projection_only"],
               "uint8_ingest": res["uint8_ingest"], "slice_rows": res.get("slice_rows"), "model": res["model"]}
        if wal is not None:
            doc["walabot_grid"] = wal
        if gen_row:
            doc["general_rows"] = gen_row
        if dnn_row is not None:
            doc["dnn_forward"] = dnn_row
        if sgan_row is not None:
            doc["sgan_train_step"] = sgan_row

        # parity gate ("mismatch counts must be 0"): any label mismatch or a decision value off by more than 1e-5 on
        # any SVM row fails the run (exit status 3, after the line is printed)
        fails = []

        def gate(name, par):
            if not par:
                return
            for k in ("label_vote_mismatch", "label_calib_mismatch", "derived_target_mismatch"):
                if par.get(k, 0) != 0:
                    fails.append("%s.%s=%d" % (name, k, par[k]))
            if "dec_ovo_max_abs_err" in par and not (par["dec_ovo_max_abs_err"] <= 1e-5):
                fails.append("%s.dec_ovo_max_abs_err=%.3g" % (name, par["dec_ovo_max_abs_err"]))

        def compact(r):
            """value / end-to-end HBM fraction / in-situ roofline fraction of the dominant kernel / parity counts of one workload row"""
            if r is None:
                return None
            c = {"v": r["value"], "e2e": r["hbm_frac_end_to_end"], "roof": r["roofline"]["frac"], "k": r["roofline"]["kernel"]}
            if r["roofline"].get("traffic") is not None and r["roofline"].get("frames_per_launch"):
                c["traffic_x"] = round(r["roofline"]["traffic"] / (r["roofline"]["algorithmic_bytes_per_frame"] * r["roofline"]["frames_per_launch"]), 3)
            par = r.get("parity")
            if par:
                c["par"] = [par["frames"], par["label_vote_mismatch"], par["label_calib_mismatch"], float("%.2g" % par["dec_ovo_max_abs_err"])]
            if r.get("gemm_roofline"):
                c["gemm"] = r["gemm_roofline"]["frac"]
            return c

        summ = {}
        for tag, r in (("g64x64x128" if grid == (64, 64, 128) else "x".join(map(str, grid)), res), ("walabot_22x31x176", wal)):
            if r is None:
                continue
            gate(tag, r["parity"])
            row = {"f32": compact(r)}
            u8r = r.get("uint8_ingest")
            if u8r:
                row["u8"] = {"v": u8r["value"], "e2e": u8r["hbm_frac_end_to_end"], "roof": u8r["roofline"]["frac"], "same_bits": u8r["identical_to_f32_ingest"]}
                if not u8r["labels_identical"]:
                    fails.append(tag + ".uint8_ingest.labels_differ")
            for key, sr in (r.get("slice_rows") or {}).items():
                gate(tag + "." + key, sr.get("parity"))
                row[key] = compact(sr)
                if key == "slice_mode" and sr.get("bound"):
                    row[key]["bound"] = "gemm"
                    row[key]["floor"] = sr["roofline"].get("frac_of_request_floor")
            so = r.get("single_observation")
            if so:
                row["one_call_us"] = [so["f32_volume_to_labels"], so["u8_volume_to_labels"], so["slice_at_sdk_target_to_labels"], so["batch_64_volumes_to_labels"]]
                if not so["same_bits_as_in_the_batch"]:
                    fails.append(tag + ".single_observation.bits_differ")
            summ[tag] = row
        if res["projection_only"]:
            summ["proj_only_configs1"] = {k: v["frac"] for k, v in res["projection_only"].items()}
        if gen_row:
            for tag, g in (("general_rows", gen_row), ("general_rows_walabot", gen_row.get("walabot_grid"))):
                if not g:
                    continue
                gate(tag, g["parity"]); gate(tag + ".f64", g["float64_mfma_path"]["parity"]); gate(tag + ".from_volumes", g["from_volumes"]["parity"])
                summ[tag] = {"v": g["value"], "mfma": g["roofline"]["frac"], "f64_v": g["float64_mfma_path"]["value"], "from_vol_v": g["from_volumes"]["value"],
                             "par": [g["parity"]["frames"], g["parity"]["label_vote_mismatch"], g["parity"]["label_calib_mismatch"],
                                     float("%.2g" % g["parity"]["dec_ovo_max_abs_err"])]}
        if dnn_row is not None:
            dp = dnn_row["parity"]
            if dp["label_mismatch"] != 0:
                fails.append("dnn.label_mismatch=%d" % dp["label_mismatch"])
            wb = dnn_row["margin_guard"]["whole_batch_check"]
            if wb["label_mismatch_vs_float32_class_chain"] != 0:
                fails.append("dnn.whole_batch.label_mismatch=%d" % wb["label_mismatch_vs_float32_class_chain"])
            ri = dnn_row["random_init"]
            if ri["parity"]["label_mismatch"] != 0:
                fails.append("dnn.random_init.label_mismatch=%d" % ri["parity"]["label_mismatch"])
            summ["dnn_configs3"] = {"v": dnn_row["value"], "v_u8": dnn_row["value_uint8_volumes"], "mfma": dnn_row["roofline"]["frac"],
                                    "random_init_v": ri["value"], "random_init_rescored": ri["rescored"],
                                    "par": [dp["frames"], dp["label_mismatch"], float("%.2g" % dp["proba_max_abs_err_vs_float64_oracle"])],
                                    "guard": [dnn_row["margin_guard"]["rescored"], dnn_row["margin_guard"]["rescored_float64"], dnn_row["margin_guard"]["rows"],
                                              dnn_row["margin_guard"]["cost_frac"]],
                                    "whole": [wb["rows"], wb["label_mismatch_vs_float32_class_chain"], wb["label_mismatch_without_guard"],
                                              float("%.2g" % wb["max_bf16_error_outside_the_gap"]), float("%.2g" % wb["half_gap"])]}
        if sgan_row is not None and "value" in sgan_row:
            summ["sgan_configs4"] = {"v": sgan_row["value"], "ms": sgan_row["ms_per_step"], "same": sgan_row["replicas_identical"],
                                     "mfma": sgan_row["roofline"]["frac"]}
        summ["parity_gate"] = "pass" if not fails else fails
        rf = res["roofline"]
        cb = res["cpu_baseline"]
        line = {
            "metric": "radar frames/s (3D-proj->SVM)", "value": res["value"], "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 volumes -> u8 codes, i8 MFMA (exact int32 dot), f64 epilogue", "data": "synthetic",
            "config": {"workload": res["config"]["workload"], "parallelism": res["config"]["parallelism"],
                       "global_frames": res["config"]["global_frames"],
                       "collective_backend": (dist.get_backend() if world > 1 else None)},
            "hbm_frac_end_to_end": res["hbm_frac_end_to_end"], "labels_crc32": res["labels_crc32"],
            "roofline": {"bound": "hbm", "achieved": rf["achieved"], "peak": rf["peak"], "unit": "GB/s", "frac": rf["frac"],
                         "traffic": rf.get("traffic"), "kernel": rf["kernel"], "avg_launch_ms": rf["avg_launch_ms"],
                         "frames_per_launch": rf["frames_per_launch"], "algorithmic_bytes_per_frame": rf["algorithmic_bytes_per_frame"]},
            "cpu_baseline": None if cb is None else {"value": cb.get("value"), "unit": "frames/s", "cores": cb.get("cores"), "kind": cb.get("kind", "port"),
                                                     "sample": (cb.get("sample") or cb.get("error") or "")[:160],
                                                     "label_mismatch_vs_gpu": cb.get("label_mismatch_vs_gpu")},
            "summary": summ,
        }
        # TWO lines: the verbose rows first, on a line of their own ({"doc": ...}; --doc-file also writes them to a file), and
        # LAST the self-contained contract object, kept under 4 KB so that whatever the driver keeps of stdout holds all of it
        # (round 4 printed one 22.6 KB object and the driver could not parse it).
        print(json.dumps({"doc": doc}))
        if a.doc_file:
            os.makedirs(os.path.dirname(os.path.abspath(a.doc_file)), exist_ok=True)
            with open(a.doc_file, "w") as f:
                json.dump({"doc": doc, "line": line}, f, indent=1)
        last = json.dumps(line)
        for drop in ("general_rows_walabot", "proj_only_configs1", "general_rows", "sgan_configs4", "dnn_configs3"):
            if len(last) < 4000:
                break
            summ.pop(drop, None)
            summ["dropped"] = summ.get("dropped", []) + [drop]
            last = json.dumps(line)
        print(last)
        sys.stdout.flush()
        gate_failed = bool(fails)
    else:
        gate_failed = False
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if gate_failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
