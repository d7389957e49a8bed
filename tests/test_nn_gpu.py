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


def test_dnn_fused_trunk_vs_fp32_cpu(rml):
    """Hand-written fused conv trunk (csrc/dnn.hip: both convolutions as bf16 MFMA implicit GEMMs) vs the fp32 CPU
    restatement of dnn.py:45-52,68-76 with shared weights."""
    dnn = importlib.import_module("radar_ml_amd.dnn")
    torch.manual_seed(7)
    cpu = dnn.define_classifier(device="cpu").eval()
    for mod in cpu.modules():                                  # non-zero biases so that they are exercised
        if isinstance(mod, torch.nn.Conv2d):
            torch.nn.init.normal_(mod.bias, 0.0, 0.1)
    gpu = copy.deepcopy(cpu).to("cuda").to(memory_format=torch.channels_last).eval()
    rng = np.random.default_rng(8)
    x = [rng.uniform(-1, 1, (37, 80, 80)).astype(np.float32) for _ in range(3)]
    with torch.no_grad():
        want = cpu.features(*[torch.from_numpy(a).unsqueeze(1) for a in x]).numpy()
        got = gpu.features_fused(*[torch.from_numpy(a).cuda() for a in x]).float().cpu().numpy()
        assert got.shape == want.shape == (37, 38400)
        err = np.abs(got - want)
        assert err.max() <= 2e-2 + 1e-2 * np.abs(want).max() and err.mean() <= 2e-3      # bf16 storage of activations
        p_want = cpu.predict([a[..., None] for a in x], autocast_dtype=None)
        p_got = gpu.forward_fused(*[torch.from_numpy(a).cuda() for a in x]).cpu().numpy()
    assert np.abs(p_got - p_want).max() < 3e-2
    # other sizes: strips of 4 / 2 conv2 rows, a partial last strip (24 rows -> 6 conv2 rows), sgan's 128x128
    for (h_, w_) in ((48, 64), (24, 16), (128, 128), (44, 36), (4, 4)):
        small = dnn.Classifier([(h_, w_, 1)] * 3, 3).eval()
        for mod in small.modules():
            if isinstance(mod, torch.nn.Conv2d):
                torch.nn.init.normal_(mod.bias, 0.0, 0.1)
        sg = copy.deepcopy(small).to("cuda").eval()
        xs = [rng.uniform(-1, 1, (5, h_, w_)).astype(np.float32) for _ in range(3)]
        with torch.no_grad():
            w = small.features(*[torch.from_numpy(a).unsqueeze(1) for a in xs]).numpy()
            g = sg.features_fused(*[torch.from_numpy(a).cuda() for a in xs]).float().cpu().numpy()
        assert g.shape == w.shape and np.abs(g - w).max() <= 2e-2 + 1e-2 * np.abs(w).max(), (h_, w_)
