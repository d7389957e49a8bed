#!/usr/bin/env python3
"""Throughput of the PyTorch-ROCm rows of the hot path (BASELINE configs[3] and configs[4]) on ONE GPU:

  dnn   volumes (Walabot arena grid) -> max-projection (HIP) -> [-1,1] scaling + bicubic 80x80 resize -> multi-view CNN
        forward in bf16 (dnn.py:45-91, 200-254); frames/s and achieved TFLOP/s (54.7 MFLOP per sample forward)
  sgan  discriminator/classifier train step (sgan.py:525-532: c_model + d_model(real) updates) on 128x128
        projections, fp16 autocast + loss scaling; samples/s

    python tools/bench_nn.py dnn  [--frames 32768] [--batch 8192]
    python tools/bench_nn.py sgan [--batch 256] [--steps 20]
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DNN_FLOP_PER_SAMPLE = 54.7e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["dnn", "sgan"])
    ap.add_argument("--frames", type=int, default=32768)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dtype", default="")
    a = ap.parse_args()
    import torch
    import torch.nn.functional as F
    import radar_ml_amd as rml
    dev = torch.device("cuda", 0)
    if a.what == "dnn":
        dnn = importlib.import_module("radar_ml_amd.dnn")
        X, Y, Z = 22, 31, 176
        B = a.frames
        bs = a.batch or 8192
        V, _ = rml.synth_volumes(B, X, Y, Z, seed=5)
        model = dnn.define_classifier(device=dev).eval()
        dt = getattr(torch, a.dtype or "bfloat16")

        def prep(p):                       # dnn.py:202-205 + 240-245 on the GPU
            p = (p - 127.5) / 127.5
            return F.interpolate(p.unsqueeze(1), size=(80, 80), mode="bicubic", align_corners=False,
                                 antialias=True).contiguous(memory_format=torch.channels_last)

        def run():
            outs = []
            with torch.no_grad(), torch.autocast("cuda", dtype=dt):
                for s in range(0, B, bs):
                    xz, yz, xy = rml.project(V[s:s + bs], mode="max")
                    outs.append(model(prep(xz), prep(yz), prep(xy)).argmax(dim=-1))
            return torch.cat(outs)

        for _ in range(2):
            lab = run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps // 4 or 1):
            lab = run()
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / (a.steps // 4 or 1)
        # forward only, resident inputs
        xs = [prep(p) for p in rml.project(V[:bs], mode="max")]
        with torch.no_grad(), torch.autocast("cuda", dtype=dt):
            for _ in range(3):
                model(*xs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                model(*xs)
            torch.cuda.synchronize()
        fwd = (time.perf_counter() - t0) / 10
        with torch.no_grad():
            for _ in range(3):
                model.forward_fused(*xs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                model.forward_fused(*xs)
            torch.cuda.synchronize()
            fused = (time.perf_counter() - t0) / 10
            for _ in range(3):
                model.features_fused(*xs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                model.features_fused(*xs)
            torch.cuda.synchronize()
            trunk = (time.perf_counter() - t0) / 10
        # the all-HIP pipeline: projection -> Pillow-exact resize (bf16 out) -> fused trunk -> dense tail
        nc = importlib.import_module("radar_ml_amd.nn_common")

        def run_hip():
            return model.predict_volumes(V, batch_size=bs).argmax(dim=-1)

        def timeit(fn, n=10):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n

        with torch.no_grad():
            t_hip = timeit(run_hip, a.steps // 4 or 1)
            lab_hip = run_hip()
            feat = rml.process_volumes(V[:bs], mode="max", scale=False)
            t_proj = timeit(lambda: rml.process_volumes(V[:bs], mode="max", scale=False))
            t_res = timeit(lambda: nc.preprocess_features(feat, (X, Y, Z), (80, 80), out_dtype="bfloat16"))
            x16 = nc.preprocess_features(feat, (X, Y, Z), (80, 80), out_dtype="bfloat16")
            t_fwd16 = timeit(lambda: model.forward_fused(*x16))
            V8 = V.to(torch.uint8)
            t_hip8 = timeit(lambda: model.predict_volumes(V8, batch_size=bs).argmax(dim=-1), a.steps // 4 or 1)
            same8 = bool(torch.equal(model.predict_volumes(V8, batch_size=bs).argmax(dim=-1), lab_hip))
            del V8
        print(json.dumps({"what": "configs[3] all-HIP from uint8 volumes", "frames": B, "frames_per_s_end_to_end": round(B / t_hip8),
                          "ms_total": round(t_hip8 * 1e3, 2), "labels_identical_to_f32_ingest": same8}))
        print(json.dumps({"what": "configs[3] all-HIP: projection + resize + fused trunk + dense tail", "frames": B, "batch": bs,
                          "frames_per_s_end_to_end": round(B / t_hip), "ms_total": round(t_hip * 1e3, 2),
                          "projection_ms": round(t_proj * 1e3, 3), "resize_ms": round(t_res * 1e3, 3),
                          "forward_ms": round(t_fwd16 * 1e3, 3), "per_batch": bs,
                          "label_agreement_with_torch_path": round(float((lab_hip == lab).float().mean()), 4)}))
        print(json.dumps({"what": "fused HIP trunk (csrc/dnn.hip) + bf16 dense tail", "batch": bs,
                          "forward_frames_per_s": round(bs / fused), "forward_TFLOPs": round(bs * DNN_FLOP_PER_SAMPLE / fused / 1e12, 1),
                          "trunk_only_frames_per_s": round(bs / trunk), "trunk_ms": round(trunk * 1e3, 3)}))
        print(json.dumps({"what": "configs[3]: projection + dnn forward", "dtype": str(dt), "frames": B, "batch": bs,
                          "frames_per_s_end_to_end": round(B / dtm), "ms_total": round(dtm * 1e3, 2),
                          "cnn_forward_frames_per_s": round(bs / fwd), "cnn_forward_TFLOPs": round(bs * DNN_FLOP_PER_SAMPLE / fwd / 1e12, 1),
                          "label_hist": torch.bincount(lab, minlength=3).tolist()}))
    else:
        sgan = importlib.import_module("radar_ml_amd.sgan")
        if os.environ.get("RML_CUDNN_BENCHMARK"):          # experiment: let MIOpen time its solvers instead of taking the heuristic pick
            torch.backends.cudnn.benchmark = bool(int(os.environ["RML_CUDNN_BENCHMARK"]))
        n = a.batch or 256
        d = sgan.define_discriminator(device=dev)
        tr = sgan.DiscriminatorTrainer(d, amp_dtype=a.dtype or "float16", ddp=False, use_graph=bool(int(os.environ.get("RML_SGAN_GRAPH", "1"))))
        g = torch.Generator(device=dev).manual_seed(0)
        x = [torch.rand((n, 128, 128), device=dev, generator=g) * 2 - 1 for _ in range(3)]
        y = torch.randint(0, 3, (n,), device=dev, generator=g)
        yr = torch.full((n, 1), 0.9, device=dev)
        for _ in range(6):
            tr.train_on_batch_c(x, y); tr.train_on_batch_d(x, yr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            lc, acc = tr.train_on_batch_c(x, y, sync=False)
            ld = tr.train_on_batch_d(x, yr, sync=False)
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / a.steps
        print(json.dumps({"what": "configs[4]: sgan discriminator step (c + d_real updates)", "dtype": a.dtype or "float16",
                          "batch": n, "ms_per_step": round(dtm * 1e3, 2), "samples_per_s": round(2 * n / dtm),
                          "c_loss": round(float(lc), 4), "d_loss": round(float(ld), 4)}))


if __name__ == "__main__":
    main()
