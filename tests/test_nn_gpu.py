"""dnn forward (bf16) and sgan discriminator step (fp16) on the GPU vs the fp32 CPU restatement with shared
random-init weights.  Parity is "unpinned" by the reference here (no weights / outputs in its tree, TensorFlow not
installable): the pins are the layer shapes and parameter counts (tests/test_nn_cpu.py)."""
import copy
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_dnn_forward_bf16_vs_fp32_cpu(rml):
    dnn = importlib.import_module("radar_ml_amd.dnn")
    torch.manual_seed(1)
    cpu = dnn.define_classifier(device="cpu")
    gpu = copy.deepcopy(cpu).to("cuda").to(memory_format=torch.channels_last)
    rng = np.random.default_rng(2)
    x = [rng.uniform(-1, 1, (256, 80, 80, 1)).astype(np.float32) for _ in range(3)]
    want = cpu.predict(x, autocast_dtype=None)
    got32 = gpu.predict(x, autocast_dtype=None)
    got16 = gpu.predict(x, autocast_dtype="bfloat16")
    assert np.abs(got32 - want).max() < 1e-4
    assert np.abs(got16 - want).max() < 3e-2                     # bf16 tolerance on probabilities
    assert (got16.argmax(1) == want.argmax(1)).mean() > 0.97


def test_preprocess_matches_pil_bicubic(rml):
    from PIL import Image
    nc = importlib.import_module("radar_ml_amd.nn_common")
    import oracle_np as O
    vol, _ = O.synth_volumes(5, 6, 22, 31, 176)
    samples = [O.project_max(v) for v in vol]
    xz, yz, xy = nc.preprocess_projections(samples, (80, 80))
    for got, idx in ((xz, 0), (yz, 1), (xy, 2)):
        for b in range(len(samples)):
            p = (samples[b][idx] - 127.5) / 127.5                 # dnn.py:202-205
            want = np.asarray(Image.fromarray(p.astype(np.float32)).resize((80, 80), resample=Image.BICUBIC))
            assert np.abs(got[b, 0].cpu().numpy() - want).max() < 2e-2


def test_sgan_step_fp16_tracks_fp32(rml):
    sgan = importlib.import_module("radar_ml_amd.sgan")
    torch.manual_seed(3)
    cpu = sgan.Discriminator(((64, 64, 1),) * 3, 3)
    gpu = copy.deepcopy(cpu).to("cuda").to(memory_format=torch.channels_last)
    tc = sgan.DiscriminatorTrainer(cpu, amp_dtype=None, ddp=False)
    tg = sgan.DiscriminatorTrainer(gpu, amp_dtype="float16", ddp=False)
    rng = np.random.default_rng(4)
    y = rng.integers(0, 3, 32)
    x = [rng.uniform(-1, 1, (32, 64, 64, 1)).astype(np.float32) for _ in range(3)]
    cpu.drop.p = gpu.drop.p = 0.0                                  # make the two runs comparable
    lc, _ = tc.train_on_batch_c(x, y); lg, _ = tg.train_on_batch_c(x, y)
    assert abs(lc - lg) < 2e-2
    dc = tc.train_on_batch_d(x, np.full((32, 1), 0.9)); dg = tg.train_on_batch_d(x, np.full((32, 1), 0.9))
    assert abs(dc - dg) < 3e-2
