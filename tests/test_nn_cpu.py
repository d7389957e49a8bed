"""dnn.py / sgan.py PyTorch modules: architecture, Keras semantics and losses on CPU (fp32)."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def mods(rml):
    return importlib.import_module("radar_ml_amd.dnn"), importlib.import_module("radar_ml_amd.sgan"), \
        importlib.import_module("radar_ml_amd.nn_common")


def test_dnn_architecture_matches_reference(mods):
    dnn, _, _ = mods
    m = dnn.define_classifier(device="cpu")
    # 3 x (640 + 18 464) + 2 457 664 + 4 160 + 195 (images/dnn_model.png)
    assert sum(p.numel() for p in m.parameters()) == 2519331
    assert m.flat_features == 20 * 20 * 96 == 38400
    x = [np.random.default_rng(i).uniform(-1, 1, (5, 80, 80, 1)).astype(np.float32) for i in range(3)]
    p = m.predict(x, autocast_dtype=None)
    assert p.shape == (5, 3) and p.dtype == np.float32
    np.testing.assert_allclose(p.sum(1), 1.0, atol=1e-6)
    assert m.predict([a[:0] for a in x]).shape == (0, 3)
    # dropout inactive at inference: deterministic
    np.testing.assert_array_equal(p, m.predict(x, autocast_dtype=None))


def test_tf_same_padding_and_flatten_order(mods):
    _, _, nc = mods
    assert nc.tf_same_pad(80, 3, 2) == (0, 1) and nc.tf_same_pad(40, 3, 2) == (0, 1)      # bottom/right only
    assert nc.tf_same_pad(81, 3, 2) == (1, 1) and nc.tf_same_pad(128, 3, 2) == (0, 1)
    conv = nc.make_same_conv(1, 2, 3, 2)
    x = torch.arange(36, dtype=torch.float32).reshape(1, 1, 6, 6)
    y = conv(x)
    assert y.shape == (1, 2, 3, 3)
    # TF semantics: output (r,c) covers input rows 2r..2r+2 with zero beyond the bottom/right edge
    w, b = conv.conv.weight.detach(), conv.conv.bias.detach()
    xp = torch.zeros(1, 1, 7, 7); xp[..., :6, :6] = x
    want = torch.stack([(xp[0, 0, 2 * r:2 * r + 3, 2 * c:2 * c + 3] * w[0, 0]).sum() + b[0] for r in range(3) for c in range(3)])
    np.testing.assert_allclose(y[0, 0].reshape(-1).detach().numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
    t = torch.arange(2 * 3 * 4 * 5).reshape(2, 3, 4, 5)       # (N,C,H,W)
    f = nc.flatten_nhwc(t)
    assert f[0, 0] == t[0, 0, 0, 0] and f[0, 1] == t[0, 1, 0, 0] and f[0, 3] == t[0, 0, 0, 1]   # (h, w, c) order


def test_sgan_discriminator_architecture_and_losses(mods):
    _, sgan, _ = mods
    d = sgan.define_discriminator(device="cpu")
    assert abs(sum(p.numel() for p in d.parameters()) - 1.86e6) < 5e3 and d.flat_features == 16 * 16 * 96
    lg = torch.randn(7, 3)
    D = sgan.custom_activation(lg)
    Z = torch.exp(lg).sum(-1, keepdim=True)
    np.testing.assert_allclose(D.numpy(), (Z / (Z + 1)).numpy(), rtol=1e-6)                 # sgan.py:125-129
    y = torch.tensor([0.9, 1.1, 0.7, 0.0, 0.2, 1.0, 0.3])
    bce = -(y * torch.log(D[:, 0]) + (1 - y) * torch.log(1 - D[:, 0])).mean()
    np.testing.assert_allclose(float(sgan.d_loss(lg, y)), float(bce), rtol=1e-5)
    yc = torch.tensor([0, 1, 2, 1, 0, 2, 1])
    np.testing.assert_allclose(float(sgan.c_loss(lg, yc)), float(-torch.log_softmax(lg, -1)[torch.arange(7), yc].mean()), rtol=1e-6)


def test_sgan_train_steps_reduce_loss(mods):
    _, sgan, _ = mods
    torch.manual_seed(0)
    d = sgan.Discriminator(((32, 32, 1),) * 3, 3)
    tr = sgan.DiscriminatorTrainer(d, amp_dtype=None, ddp=False)
    rng = np.random.default_rng(0)
    y = rng.integers(0, 3, 48)
    x = [((rng.uniform(-1, 1, (48, 32, 32, 1)) * 0.1) + (y[:, None, None, None] - 1) * 0.5).astype(np.float32) for _ in range(3)]
    l0, _ = tr.train_on_batch_c(x, y)
    for _ in range(30):
        l, acc = tr.train_on_batch_c(x, y)
    assert l < l0 * 0.8 and acc > 0.6
    d0 = tr.train_on_batch_d(x, np.full((48, 1), 0.9))
    for _ in range(20):
        dl = tr.train_on_batch_d(x, np.full((48, 1), 0.9))
    assert dl < d0
    p = tr.predict(x)
    assert p.shape == (48, 3) and abs(p.sum(1) - 1).max() < 1e-5


def _ddp_worker(rank, world, port, q, mode=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import radar_ml_amd  # noqa
    sgan = importlib.import_module("radar_ml_amd.sgan")
    torch.manual_seed(rank)                                    # DIFFERENT initial weights: the trainer must broadcast rank 0's
    d = sgan.Discriminator(((16, 16, 1),) * 3, 3).to(memory_format=torch.channels_last)     # as define_discriminator lays it out
    tr = sgan.DiscriminatorTrainer(d, amp_dtype=None, ddp=mode)
    if mode is None:
        assert all(p.grad.stride() == p.stride() for p in d.parameters())                 # flat-bucket views follow the parameters' layout     # None: picks the flat-bucket all-reduce up from the process group
    rng = np.random.default_rng(100 + rank)                    # each rank its own shard of the batch
    x = [rng.uniform(-1, 1, (8, 16, 16, 1)).astype(np.float32) for _ in range(3)]
    tr.train_on_batch_c(x, rng.integers(0, 3, 8))
    tr.train_on_batch_d(x, np.full((8, 1), 0.9))
    flat = torch.cat([p.detach().reshape(-1) for p in d.parameters()])
    q.put((rank, flat.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(mode):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_sgan_data_parallel_gloo_two_ranks_keep_replicas_in_sync():
    """configs[4] with N > 1 on CPU (gloo, world size 2): the flat-bucket gradient all-reduce (the default) keeps the replicas
    bit-identical from different initial weights, and lands where torch's DistributedDataParallel lands."""
    flat = _run_two_ranks(None)
    np.testing.assert_array_equal(flat[0], flat[1])     # all-reduced gradients -> identical replicas
    ddp = _run_two_ranks("torch")
    np.testing.assert_array_equal(ddp[0], ddp[1])
    assert np.abs(flat[0] - ddp[0]).max() < 1e-6        # the same mean gradient, summed in a different order


def test_dnn_module_matches_numpy_oracle(mods):
    """The PyTorch module against the NumPy restatement of the Keras layers (oracle_np.dnn_forward, dnn.py:45-91):
    same weights, odd and even sizes ('same' padding differs between them), float32 vs float64."""
    import oracle_np as O
    dnn, _, _ = mods
    rng = np.random.default_rng(3)
    for (h, w) in ((80, 80), (22, 31), (9, 12)):
        torch.manual_seed(h)
        m = dnn.Classifier([(h, w, 1)] * 3, 3).eval()
        for mod in m.modules():
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear)):
                torch.nn.init.normal_(mod.bias, 0.0, 0.1)
        convs, dense = m.keras_weights()
        assert convs[0][0].shape == (3, 3, 1, 64) and convs[0][2].shape == (3, 3, 64, 32) and dense[0][0].shape[1] == 64
        x = [rng.uniform(-1, 1, (4, h, w)).astype(np.float32) for _ in range(3)]
        want_f = O.dnn_conv_features(*x, convs)
        with torch.no_grad():
            got_f = m.features(*[torch.from_numpy(a).unsqueeze(1) for a in x]).numpy()
        assert got_f.shape == want_f.shape
        assert np.abs(got_f - want_f).max() < 1e-4
        want_p = O.dnn_forward(*x, convs, dense)
        got_p = m.predict([a[..., None] for a in x], autocast_dtype=None)
        assert np.abs(got_p - want_p).max() < 1e-5


def test_bn_lrelu_pad_cpu_fallback_and_branch_equivalence(mods):
    """nn_common.bn_lrelu_pad on CPU tensors is the plain PyTorch layers (the fused HIP op needs a GPU and half
    precision), including the 'bias of the producing convolution' argument; Discriminator._branch falls back to the
    Sequential of the reference's layers."""
    import torch.nn.functional as F
    _, sgan, nc = mods
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm2d(16, eps=1e-3, momentum=0.01).train()
    ref = torch.nn.BatchNorm2d(16, eps=1e-3, momentum=0.01).train()
    ref.load_state_dict(bn.state_dict())
    x = torch.randn(4, 16, 6, 8)
    b = torch.randn(16)
    got = nc.bn_lrelu_pad(x, bn, 0.2, pad=1, conv_bias=b)
    want = F.pad(F.leaky_relu(ref(x + b.reshape(1, -1, 1, 1)), 0.2), (0, 1, 0, 1))
    assert got.shape == (4, 16, 7, 9) and torch.allclose(got, want, atol=1e-6)
    assert torch.allclose(bn.running_mean, ref.running_mean) and torch.allclose(bn.running_var, ref.running_var)
    d = sgan.Discriminator(((16, 16, 1),) * 3, 3).train()
    xin = torch.randn(5, 1, 16, 16)
    torch.manual_seed(1); a = d._branch(xin, d.branches[0])
    d2 = sgan.Discriminator(((16, 16, 1),) * 3, 3).train()
    d2.load_state_dict(d.state_dict())
    # same statistics update and output as running the Sequential directly
    b_ = d2.branches[0](xin)
    assert a.shape == b_.shape == (5, 32, 2, 2)


def test_gpu_only_entry_points_fail_loudly_without_a_gpu(mods):
    """No CPU fallback for the HIP ops: the host wrappers raise instead of silently computing elsewhere."""
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    import radar_ml_amd as rml
    _, _, nc = mods
    with pytest.raises(ValueError):
        nc.resize_bicubic(torch.zeros((2, 8, 8)), (4, 4))                 # CPU tensor
    with pytest.raises(rml.RadarMLError):
        rml.KernelMatrix(np.zeros((4, 8)), gamma=0.1)
    with pytest.raises(rml.RadarMLError):
        rml.process_volumes(np.zeros((1, 4, 4, 8), np.uint8))


def test_keras_semantics_known_answers_derived_by_hand(mods):
    """Known answers that need no TensorFlow: derived by hand from the TF/Keras definitions cited in SURVEY.md §7.

    TF padding='same', stride s, kernel k on size n: out = ceil(n/s); total pad = max((out-1)*s + k - n, 0);
    pad_before = total // 2, pad_after = total - pad_before.  For n = 4, k = 3, s = 2: out = 2, total = 1 -> (0, 1): the zero
    row/column goes to the BOTTOM/RIGHT.  An all-ones 4x4 image through an all-ones 3x3 kernel then gives
        [[9, 6], [6, 4]]        (window rows/cols {0,1,2} and {2,3,pad})
    whereas symmetric padding 1 (PyTorch's padding=1) would give [[4, 6], [6, 9]].  For n = 5: out = 3, total = 2 -> (1, 1):
        [[4, 6, 4], [6, 9, 6], [4, 6, 4]].
    Keras Flatten on channels_last (N,H,W,C) enumerates (h, w, c) with c fastest."""
    _, sgan, nc = mods
    conv = nc.make_same_conv(1, 1, 3, 2)
    with torch.no_grad():
        conv.conv.weight.fill_(1.0); conv.conv.bias.zero_()
        np.testing.assert_array_equal(conv(torch.ones(1, 1, 4, 4))[0, 0].numpy(), [[9, 6], [6, 4]])
        np.testing.assert_array_equal(conv(torch.ones(1, 1, 5, 5))[0, 0].numpy(), [[4, 6, 4], [6, 9, 6], [4, 6, 4]])
        # a one-hot image: tap (ky, kx) of output (r, c) reads input (2r + ky, 2c + kx) -- no shift from a top/left pad
        w = torch.arange(9, dtype=torch.float32).reshape(1, 1, 3, 3)
        conv.conv.weight.copy_(w)
        x = torch.zeros(1, 1, 6, 6); x[0, 0, 3, 4] = 1.0
        y = conv(x)[0, 0]
        want = torch.zeros(3, 3)
        for r in range(3):
            for c in range(3):
                ky, kx = 3 - 2 * r, 4 - 2 * c
                if 0 <= ky < 3 and 0 <= kx < 3:
                    want[r, c] = w[0, 0, ky, kx]
        np.testing.assert_array_equal(y.numpy(), want.numpy())
    t = torch.zeros(1, 2, 2, 2)
    for c in range(2):
        for h in range(2):
            for w_ in range(2):
                t[0, c, h, w_] = 100 * h + 10 * w_ + c
    np.testing.assert_array_equal(nc.flatten_nhwc(t)[0].numpy(), [0, 1, 10, 11, 100, 101, 110, 111])
    # LeakyReLU(0.2), BatchNorm(momentum 0.99 -> torch 0.01, eps 1e-3), Adam(2e-4, beta1 0.5, eps 1e-7): sgan.py:137-158, 206
    d = sgan.Discriminator(((16, 16, 1),) * 3, 3)
    bns = [m for m in d.modules() if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d))]
    assert len(bns) == 11 and all(abs(b.momentum - 0.01) < 1e-12 and abs(b.eps - 1e-3) < 1e-12 for b in bns)
    assert all(abs(m.negative_slope - 0.2) < 1e-12 for m in d.modules() if isinstance(m, torch.nn.LeakyReLU))
    tr = sgan.DiscriminatorTrainer(d, amp_dtype=None, ddp=False)
    for opt in (tr.opt_c, tr.opt_d):
        g = opt.param_groups[0]
        assert g["lr"] == 2e-4 and g["betas"] == (0.5, 0.999) and g["eps"] == 1e-7


def test_parameter_counts_match_the_model_summaries_exactly(mods):
    """images/dnn_model.png, images/sgan_d_model.png / sgan_c_model.png (layer shapes) -> Keras parameter counts, derived in
    SURVEY.md §8 a-9 / a-10.  Keras counts BatchNorm's moving mean / variance as (non-trainable) parameters: in PyTorch they
    are buffers."""
    dnn, sgan, _ = mods
    m = dnn.define_classifier(device="cpu")
    per_branch = (3 * 3 * 1 * 64 + 64) + (3 * 3 * 64 * 32 + 32)
    assert per_branch == 640 + 18464
    assert sum(p.numel() for p in m.parameters()) == 3 * per_branch + (38400 * 64 + 64) + (64 * 64 + 64) + (64 * 3 + 3) == 2519331
    shapes = [tuple(p.shape) for p in m.branches[0].parameters()]
    assert shapes == [(64, 1, 3, 3), (64,), (32, 64, 3, 3), (32,)]
    d = sgan.define_discriminator(device="cpu")
    conv_bn = lambda cin, cout: (9 * cin * cout + cout) + 4 * cout          # kernel + bias + (gamma, beta, mean, var)
    branch = conv_bn(1, 128) + conv_bn(128, 64) + conv_bn(64, 32)
    assert branch == 94432
    keras_total = 3 * branch + (24576 * 64 + 64) + 4 * 64 + (64 * 64 + 64) + 4 * 64 + (64 * 3 + 3)
    assert keras_total == 1861091
    n_param = sum(p.numel() for p in d.parameters())
    n_stat = sum(b.numel() for n, b in d.named_buffers() if not n.endswith("num_batches_tracked"))
    assert n_param + n_stat == keras_total and n_stat == 2 * (3 * (128 + 64 + 32) + 64 + 64)
    x = [torch.zeros(2, 1, 128, 128) for _ in range(3)]
    d.eval()
    with torch.no_grad():
        h = d.branches[0](x[0])
    assert tuple(h.shape) == (2, 32, 16, 16) and d(*x).shape == (2, 3)          # 128 -> 64 -> 32 -> 16 per branch


def test_keras_layout_weights_round_trip_and_flat_list(mods):
    """set_keras_weights is the inverse of keras_weights (conv kernels (kh,kw,cin,cout), dense kernels (in,out)); the flat
    ``model.get_weights()`` list loads in both layer orders; a wrong shape is refused.  The arrays are also run through the
    NumPy restatement of the Keras layers (oracle_np.dnn_forward), which reads them in Keras layout on its own."""
    import oracle_np as O
    dnn, _, _ = mods
    rng = np.random.default_rng(12)
    h, w = 12, 10
    torch.manual_seed(1)
    src = dnn.Classifier([(h, w, 1)] * 3, 3).eval()
    for mod in src.modules():
        if isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear)):
            torch.nn.init.normal_(mod.bias, 0.0, 0.1)
    convs, dense = src.keras_weights()
    x = [rng.uniform(-1, 1, (5, h, w, 1)).astype(np.float32) for _ in range(3)]
    want = src.predict(x, autocast_dtype=None)
    torch.manual_seed(2)
    dst = dnn.Classifier([(h, w, 1)] * 3, 3).eval()
    assert np.abs(dst.predict(x, autocast_dtype=None) - want).max() > 1e-3          # different weights to start with
    dst.set_keras_weights(convs, dense)
    np.testing.assert_array_equal(dst.predict(x, autocast_dtype=None), want)
    assert np.abs(want - O.dnn_forward(*[a[..., 0] for a in x], convs, dense)).max() < 1e-5
    flat_depth = [a for b in convs for a in b[:2]] + [a for b in convs for a in b[2:]] + [a for kb in dense for a in kb]
    flat_branch = [a for b in convs for a in b] + [a for kb in dense for a in kb]
    for flat, order in ((flat_depth, "depth"), (flat_branch, "branch")):
        torch.manual_seed(3)
        m = dnn.Classifier([(h, w, 1)] * 3, 3).eval().set_keras_weight_list(flat, order=order)
        np.testing.assert_array_equal(m.predict(x, autocast_dtype=None), want)
    with pytest.raises(ValueError):
        dst.set_keras_weights([(c[0].transpose(3, 2, 0, 1), c[1], c[2], c[3]) for c in convs], dense)       # torch layout: refused
    with pytest.raises(ValueError):
        dst.set_keras_weight_list(flat_depth[:-1])


def test_keras_layout_known_answer_through_the_importer(mods):
    """Hand-derived answer that only holds if (kh, kw, cin, cout) / (in, out) are read as Keras writes them.
    4x4 all-ones planes; conv1 kernel (3,3,1,64) = 1 on channel 0 only; TF 'same' stride 2 gives channel 0 =
    [[9, 6], [6, 4]] (pad bottom/right).  conv2 kernel (3,3,64,32): k2[ky, kx, 0, 0] = 10 ky + kx + 1, everything else 0:
    the single 1x1 output of channel 0 reads taps (0..1, 0..1) of the 2x2 image plus the zero pad:
    1*9 + 2*6 + 11*6 + 12*4 = 135.  Dense kernels (in, out): dense[0][0][0, 1] = 2 routes feature 0 (branch xz, channel 0) to
    unit 1 -> 270; dense_1 passes unit 1 to unit 2; dense_2 kernel [2, :] = (0, 1, 0) and bias (0, 0, 269): logits (0, 270, 269)."""
    dnn, _, _ = mods
    m = dnn.Classifier([(4, 4, 1)] * 3, 3).eval()
    k1 = np.zeros((3, 3, 1, 64)); k1[:, :, 0, 0] = 1.0
    k2 = np.zeros((3, 3, 64, 32))
    for ky in range(3):
        for kx in range(3):
            k2[ky, kx, 0, 0] = 10 * ky + kx + 1
    z1, z2 = np.zeros(64), np.zeros(32)
    convs = [(k1, z1, k2, z2), (np.zeros_like(k1), z1, np.zeros_like(k2), z2), (np.zeros_like(k1), z1, np.zeros_like(k2), z2)]
    d0 = np.zeros((96, 64)); d0[0, 1] = 2.0
    d1 = np.zeros((64, 64)); d1[1, 2] = 1.0
    d2 = np.zeros((64, 3)); d2[2, 1] = 1.0
    m.set_keras_weights(convs, [(d0, np.zeros(64)), (d1, np.zeros(64)), (d2, np.array([0.0, 0.0, 269.0]))])
    x = [torch.ones(1, 1, 4, 4) for _ in range(3)]
    with torch.no_grad():
        f = m.features(*x)[0].numpy()
        lg = m.logits(*x)[0].numpy()
    assert f.shape == (96,) and f[0] == 135.0 and np.count_nonzero(f) == 1
    np.testing.assert_array_equal(lg, [0.0, 270.0, 269.0])
    p = m.predict([np.ones((1, 4, 4, 1), np.float32)] * 3, autocast_dtype=None)[0]
    assert abs(p[1] - 1.0 / (1.0 + np.exp(-1.0))) < 1e-6 and p[0] < 1e-30
    loss, acc = m.evaluate([np.ones((2, 4, 4, 1), np.float32)] * 3, [1, 2])
    assert acc == 0.5 and abs(loss - 0.5 * (np.log1p(np.exp(-1.0)) + np.log1p(np.exp(1.0)))) < 1e-6


def test_discriminator_keras_weights_evaluate_and_class_weight(mods):
    """sgan: Keras-layout export / import with the BatchNorm moving statistics (gamma, beta, moving_mean, moving_variance),
    c_model.evaluate (sgan.py:491), and class_weight on the d update (sgan.py:529-530) as Keras turns it into sample weights."""
    _, sgan, _ = mods
    torch.manual_seed(0)
    src = sgan.Discriminator(((16, 16, 1),) * 3, 3)
    tr = sgan.DiscriminatorTrainer(src, amp_dtype=None, ddp=False)
    rng = np.random.default_rng(4)
    x = [rng.uniform(-1, 1, (8, 16, 16, 1)).astype(np.float32) for _ in range(3)]
    y = rng.integers(0, 3, 8)
    for _ in range(3):                                   # move the BatchNorm statistics and the parameters off their defaults
        tr.train_on_batch_c(x, y)
    branches, dense = src.keras_weights()
    assert branches[0][0][0].shape == (3, 3, 1, 128) and len(branches[0][0]) == 6 and dense[0][0].shape == (384, 64) and len(dense[2]) == 2
    want = tr.predict(x)
    torch.manual_seed(5)
    dst = sgan.Discriminator(((16, 16, 1),) * 3, 3).set_keras_weights(branches, dense)
    tr2 = sgan.DiscriminatorTrainer(dst, amp_dtype=None, ddp=False)
    np.testing.assert_array_equal(tr2.predict(x), want)
    loss, acc = tr2.evaluate(x, y)
    p = want.astype(np.float64)
    assert abs(loss + np.log(np.clip(p[np.arange(8), y], 1e-7, 1 - 1e-7)).mean()) < 1e-12
    assert acc == float((p.argmax(1) == y).mean())
    with pytest.raises(ValueError):
        dst.set_keras_weights(branches[:2], dense)
    # class_weight -> sample weights: int(y) truncates the smoothed labels (0.7..1.2 -> class 0 or 1)
    yr = np.array([[0.71], [0.99], [1.0], [1.19], [0.3], [1.05], [0.85], [1.1]])
    w = sgan.class_weight_to_sample_weight(yr, {0: 1.0, 1: 2.5, 2: 4.0})
    np.testing.assert_array_equal(w, [1.0, 1.0, 2.5, 2.5, 1.0, 2.5, 1.0, 2.5])
    torch.manual_seed(7)
    a = sgan.Discriminator(((16, 16, 1),) * 3, 3); b = sgan.Discriminator(((16, 16, 1),) * 3, 3)
    b.load_state_dict(a.state_dict())
    ta = sgan.DiscriminatorTrainer(a, amp_dtype=None, ddp=False); tb = sgan.DiscriminatorTrainer(b, amp_dtype=None, ddp=False)
    a.drop.p = b.drop.p = 0.0                            # dropout off: the two updates must be identical
    la = ta.train_on_batch_d(x, yr, class_weight={0: 1.0, 1: 2.5, 2: 4.0})
    lb = tb.train_on_batch_d(x, yr, sample_weight=w)
    assert la == lb
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)
    with pytest.raises(ValueError):
        ta.train_on_batch_d(x, yr, sample_weight=w, class_weight={0: 1.0})
