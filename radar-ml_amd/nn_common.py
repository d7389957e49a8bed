"""Keras-semantics helpers shared by the dnn / sgan PyTorch-ROCm modules."""
import numpy as np


def _torch():
    import torch
    return torch


def tf_same_pad(size, kernel, stride):
    """TensorFlow padding='same': (before, after).  On even sizes with stride 2 and k=3 this is (0, 1):
    bottom/right only, unlike PyTorch's symmetric padding."""
    out = -(-size // stride)
    total = max((out - 1) * stride + kernel - size, 0)
    return total // 2, total - total // 2


def make_same_conv(in_ch, out_ch, kernel, stride):
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    class _Conv(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(in_ch, out_ch, kernel, stride=stride, padding=0, bias=True)
            self.kernel, self.stride = kernel, stride

        def forward(self, x):
            ph = tf_same_pad(x.shape[-2], self.kernel, self.stride)
            pw = tf_same_pad(x.shape[-1], self.kernel, self.stride)
            return self.conv(F.pad(x, (pw[0], pw[1], ph[0], ph[1])))

    return _Conv()


def to_nchw(a, device, dtype=None):
    """Keras feeds (N,H,W,1) or (N,H,W) numpy arrays; return an (N,1,H,W) channels_last tensor on device."""
    torch = _torch()
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    if t.ndim == 4 and t.shape[-1] == 1:
        t = t[..., 0]
    if t.ndim != 3:
        raise ValueError("expected (N,H,W) or (N,H,W,1), got %s" % (tuple(t.shape),))
    t = t.to(device=device, dtype=dtype or torch.float32).unsqueeze(1)
    return t.contiguous(memory_format=torch.channels_last)


def flatten_nhwc(x):
    """Keras Flatten on an NHWC tensor: (N,C,H,W) -> (N, H*W*C) in (h, w, c) order, so that Dense kernels
    trained in Keras keep their row order."""
    return x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)


def resize_bicubic(planes, out_hw, shape=None, scale=True, out_dtype="float32"):
    """``Image.fromarray(p).resize((out_w, out_h), Image.BICUBIC)`` for a batch of float32 planes on the GPU
    (csrc/resize.hip, bit-identical to Pillow), preceded by the reference's [-1,1] scaling ``(p-127.5)/127.5``
    (dnn.py:202-205) when ``scale``.  ``planes``: (N,H,W) CUDA float32, or an (N, H*W) view into wider rows (e.g.
    one projection inside feature rows [xz|yz|xy]) together with ``shape=(H, W)``.  Returns (N, out_h, out_w) in
    float32 or bfloat16 (the conv trunk's operand type)."""
    torch = _torch()
    from . import _lib
    lib = _lib.load()
    if planes.dtype != torch.float32 or not planes.is_cuda:
        raise ValueError("resize_bicubic: CUDA float32 planes expected")
    if shape is None:
        if planes.ndim != 3:
            raise ValueError("resize_bicubic: (N,H,W) planes expected (or pass shape=(H, W))")
        H, W = int(planes.shape[1]), int(planes.shape[2])
        flat = planes.reshape(planes.shape[0], H * W) if planes.is_contiguous() else planes.contiguous().reshape(planes.shape[0], H * W)
    else:
        H, W = int(shape[0]), int(shape[1])
        flat = planes
        if flat.ndim != 2 or flat.shape[1] != H * W:
            raise ValueError("resize_bicubic: an (N, %d) view expected for shape %s" % (H * W, (H, W)))
    if flat.shape[0] > 1 and flat.stride(1) != 1:
        flat = flat.contiguous()
    n = int(flat.shape[0])
    stride = int(flat.stride(0)) if n > 1 else H * W
    oh, ow = int(out_hw[0]), int(out_hw[1])
    dt = getattr(torch, out_dtype) if isinstance(out_dtype, str) else out_dtype
    if dt not in (torch.float32, torch.bfloat16):
        raise ValueError("resize_bicubic: out_dtype must be float32 or bfloat16")
    out = torch.empty((n, oh, ow), dtype=dt, device=flat.device)
    with torch.cuda.device(flat.device):
        _lib.check(lib.rml_resize_bicubic(_lib.context(flat.device), _lib.ptr(flat), stride, n, H, W, oh, ow,
                                          127.5 if scale else 0.0, 127.5 if scale else 0.0, _lib.ptr(out),
                                          1 if dt == torch.bfloat16 else 0, _lib.stream_ptr(flat.device)),
                   "rml_resize_bicubic")
    return out


def preprocess_projections(samples, rescale, device=None, out_dtype="float32"):
    """dnn.py:200-254 / sgan.py:636-690 on the GPU: scale [0,255] -> [-1,1] ((p-127.5)/127.5), resize every
    projection to ``rescale`` (PIL's (width, height) order, square in the reference) with Pillow's bicubic resize,
    return three (N,1,H,W) tensors in (xz, yz, xy) order."""
    torch = _torch()
    dev = device or torch.device("cuda", torch.cuda.current_device())
    outs = []
    for i in range(3):
        p = torch.from_numpy(np.stack([np.asarray(s[i], dtype=np.float32) for s in samples])).to(dev)
        outs.append(resize_bicubic(p, (rescale[1], rescale[0]), out_dtype=out_dtype).unsqueeze(1))
    return outs


def preprocess_features(feat, grid, rescale, out_dtype="bfloat16"):
    """The same preprocessing straight from unscaled feature rows [xz | yz | xy] of ``process_volumes`` (one kernel
    per projection, no copies): returns (xz, yz, xy) as (N, H, W) tensors."""
    X, Y, Z = grid
    shapes = ((X, Z), (Y, Z), (X, Y))
    outs, off = [], 0
    for (h, w) in shapes:
        outs.append(resize_bicubic(feat[:, off:off + h * w], (rescale[1], rescale[0]), shape=(h, w), out_dtype=out_dtype))
        off += h * w
    return outs


def preprocess_supported(grid, rescale):
    """True when the fused preprocessing kernel (csrc/preprocess.hip) takes this (X, Y, Z) grid and (width, height) target."""
    from . import _lib
    X, Y, Z = (int(v) for v in grid)
    return bool(_lib.load().rml_dnn_preprocess_supported(X, Y, Z, int(rescale[1]), int(rescale[0])))


def preprocess_rows(grid, rescale, feat=None, codes=None, flags=None):
    """The CNN preprocessing of dnn.py:200-254 for the bf16 conv trunk in ONE launch (csrc/preprocess.hip): float32 feature rows
    [xz | yz | xy] (``feat``) and / or the biased uint8 code rows of ``process_volumes(codes=...)`` (``codes``; with both,
    ``flags[b] != 0`` selects the code row) -> (xz, yz, xy) as (N, H, W) bfloat16.  Pillow's windows and weights applied in
    float32: within one bf16 ulp of ``resize_bicubic(..., out_dtype='bfloat16')`` (the Pillow-exact kernel), not bit-identical."""
    torch = _torch()
    from . import _lib
    lib = _lib.load()
    X, Y, Z = (int(v) for v in grid)
    src = feat if feat is not None else codes
    if src is None:
        raise ValueError("preprocess_rows: feat and / or codes expected")
    n, dev = int(src.shape[0]), src.device
    oh, ow = int(rescale[1]), int(rescale[0])
    outs = [torch.empty((n, oh, ow), dtype=torch.bfloat16, device=dev) for _ in range(3)]
    with torch.cuda.device(dev):
        _lib.check(lib.rml_dnn_preprocess_rows(_lib.context(dev), _lib.ptr(feat), int(feat.stride(0)) if feat is not None else 0,
                                               _lib.ptr(codes), int(codes.stride(0)) if codes is not None else 0, _lib.ptr(flags),
                                               n, X, Y, Z, oh, ow, _lib.ptr(outs[0]), _lib.ptr(outs[1]), _lib.ptr(outs[2]),
                                               _lib.stream_ptr(dev)), "rml_dnn_preprocess_rows")
    return outs


def preprocess_volumes(volumes, rescale, mode="max", ijk=None):
    """(N, X, Y, Z) volumes (float32 or uint8, CUDA) -> the trunk's three bf16 inputs (N, H, W): projection into uint8 code rows
    (no float rows through HBM when the projections are integers 0..255 -- radar magnitudes, common.py:30-31; rows that are not
    take a device-predicated float pass), then the fused scaling + bicubic resize (``rml_dnn_preprocess_volumes``).  Finite
    projections assumed on the float pass (its aligned windows touch neighbouring rows under zero weights: 0 * NaN is NaN where
    Pillow would not have read): ``Classifier.predict_volumes`` sends mode "max_nan" through the Pillow-exact resize instead."""
    torch = _torch()
    from . import _lib, common
    lib = _lib.load()
    v, vdt = common._as_device_volumes(volumes)
    n, X, Y, Z = (int(t) for t in v.shape)
    dev = v.device
    m = _lib.MODES[mode]
    ijk_t = None
    if m == _lib.MODE_SLICE:
        if ijk is None:
            raise ValueError("preprocess_volumes: mode='slice' needs ijk")
        ijk_t, T = common._slice_indices(ijk, n, X, Y, Z, dev)
        if T != 1:
            raise ValueError("preprocess_volumes: one (i,j,k) per frame")
    D = common.feature_len(X, Y, Z)
    ldq = (D + 127) // 128 * 128
    oh, ow = int(rescale[1]), int(rescale[0])
    codes = torch.empty((n, ldq), dtype=torch.uint8, device=dev)
    flags = torch.empty((n + 1,), dtype=torch.int32, device=dev)
    rows = torch.empty((n, D), dtype=torch.float32, device=dev) if vdt == _lib.VOL_F32 else None
    outs = [torch.empty((n, oh, ow), dtype=torch.bfloat16, device=dev) for _ in range(3)]
    with torch.cuda.device(dev):
        _lib.check(lib.rml_dnn_preprocess_volumes(_lib.context(dev), _lib.ptr(v), vdt, n, X, Y, Z, m, _lib.ptr(ijk_t), _lib.ptr(codes), ldq,
                                                  _lib.ptr(flags), _lib.ptr(rows), D, oh, ow, _lib.ptr(outs[0]), _lib.ptr(outs[1]),
                                                  _lib.ptr(outs[2]), _lib.stream_ptr(dev)), "rml_dnn_preprocess_volumes")
    return outs


class fused_step_scope:
    """Collects the small side effects of the fused SGAN layers over one forward pass and applies them as multi-tensor launches
    on exit: ``num_batches_tracked += 1`` and the running-mean correction for the dropped convolution bias of every
    ``bn_lrelu_pad`` / ``conv1_bn_lrelu_pad`` layer (18 launches of ~4.5 us per forward of the three-branch discriminator ->
    2; the step is GPU-bound and a third of its 480 launches are kernels of that size, `tools/step_gaps.py`).
    ``bias_grads=False``: the backward of those layers hands back no gradient for the dropped convolution bias instead of a
    tensor of zeros (it IS exactly zero; Adam leaves a parameter without gradient alone, which is what a zero gradient does to
    it) -- torch's DistributedDataParallel needs the zeros, nothing else does."""
    active = None

    def __init__(self, bias_grads=True):
        self.bias_grads = bool(bias_grads)
        self.items = []                 # (bn, conv_bias or None)

    def __enter__(self):
        self.prev = fused_step_scope.active
        fused_step_scope.active = self
        return self

    def __exit__(self, et, ev, tb):
        fused_step_scope.active = self.prev
        if et is None:
            self.flush()
        return False

    def flush(self):
        torch = _torch()
        if not self.items:
            return
        with torch.no_grad():
            torch._foreach_add_([bn.num_batches_tracked for bn, _ in self.items], 1)
            by_m = {}
            for bn, b in self.items:
                if b is not None:
                    by_m.setdefault((float(bn.momentum), bn.running_mean.dtype), []).append((bn.running_mean, b.detach()))
            for (m, dt), pairs in by_m.items():
                torch._foreach_add_([rm for rm, _ in pairs], [b if b.dtype == dt else b.to(dt) for _, b in pairs], alpha=m)
        self.items = []


def _bn_side_effects(bn, conv_bias):
    """after a fused training-mode batch norm: the batch counter, and the running mean of conv(x) + bias (the kernel tracked the
    mean of the bias-free activations; eval mode and the unfused layers normalise conv(x) + bias, whose mean is larger by exactly
    the bias: running_mean += momentum * bias keeps both in step) -- now, or at the end of the enclosing fused_step_scope"""
    torch = _torch()
    sc = fused_step_scope.active
    if sc is not None:
        sc.items.append((bn, conv_bias))
        return
    with torch.no_grad():
        bn.num_batches_tracked += 1
        if conv_bias is not None:
            bn.running_mean.add_(conv_bias.detach().to(bn.running_mean.dtype), alpha=float(bn.momentum))


def cast_all(dtype, *tensors):
    """``[t.to(dtype) for t in tensors]`` as ONE multi-tensor launch forward and one backward (autocast casts every weight where
    it is used: a ~4.5 us launch per parameter in each direction)."""
    torch = _torch()
    if getattr(cast_all, "_cls", None) is None:
        class CastAll(torch.autograd.Function):
            @staticmethod
            def forward(ctx, dtype, *ws):
                outs = [torch.empty_like(w, dtype=dtype) for w in ws]          # empty_like keeps channels_last kernels as they are
                torch._foreach_copy_(outs, list(ws))
                ctx.src_dtypes = [w.dtype for w in ws]
                return tuple(outs)

            @staticmethod
            def backward(ctx, *gs):
                idx = [i for i, g in enumerate(gs) if g is not None]
                outs = [torch.empty_like(gs[i], dtype=ctx.src_dtypes[i]) for i in idx]
                if idx:
                    torch._foreach_copy_(outs, [gs[i] for i in idx])
                res = [None] * len(gs)
                for i, o in zip(idx, outs):
                    res[i] = o
                return (None, *res)
        cast_all._cls = CastAll
    return cast_all._cls.apply(dtype, *tensors)


def _bn_lrelu_pad_function():
    """torch.autograd.Function around csrc/bnact.hip (built lazily: torch is imported on first use)."""
    torch = _torch()
    if getattr(_bn_lrelu_pad_function, "_cls", None) is not None:
        return _bn_lrelu_pad_function._cls
    from . import _lib

    class BnLReluPad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, slope, pad, conv_bias):
            # conv_bias: the bias of the convolution that produced x WITHOUT it.  Batch norm subtracts the batch mean, so
            # a per-channel constant in front of it changes nothing and its gradient is exactly zero: it is not added, and
            # backward hands back zeros (the parameter still takes part in the graph, which DDP requires).
            lib = _lib.load()
            n, c, h, w = x.shape
            dev = x.device
            y = torch.empty((n, c, h + pad, w + pad), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
            mean = torch.empty((c,), dtype=torch.float32, device=dev)
            rstd = torch.empty((c,), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                hctx = _lib.context(dev)
                ws = torch.empty((int(lib.rml_bn_workspace_floats(hctx, c)) + 2 * c,), dtype=torch.float32, device=dev)
                _lib.check(lib.rml_bn_lrelu_pad_forward(
                    hctx, _lib.ptr(x), 1 if x.dtype == torch.bfloat16 else 0, n, h, w, c, pad, pad, _lib.ptr(gamma), _lib.ptr(beta),
                    float(eps), float(momentum), float(slope), _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(mean),
                    _lib.ptr(rstd), _lib.ptr(ws), _lib.ptr(y), _lib.stream_ptr(dev)), "rml_bn_lrelu_pad_forward")
            ctx.save_for_backward(x, gamma, beta, mean, rstd)
            ctx.meta = (float(slope), int(pad))
            sc = fused_step_scope.active
            ctx.bias_like = conv_bias if (sc is None or sc.bias_grads) else None
            return y

        @staticmethod
        def backward(ctx, dy):
            lib = _lib.load()
            x, gamma, beta, mean, rstd = ctx.saved_tensors
            slope, pad = ctx.meta
            n, c, h, w = x.shape
            dev = x.device
            if dy.dtype != x.dtype or not dy.is_contiguous(memory_format=torch.channels_last):
                dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
            dx = torch.empty_like(x, memory_format=torch.channels_last)
            dgamma = torch.empty((c,), dtype=torch.float32, device=dev)
            dbeta = torch.empty((c,), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                hctx = _lib.context(dev)
                ws = torch.empty((int(lib.rml_bn_workspace_floats(hctx, c)) + 2 * c,), dtype=torch.float32, device=dev)
                _lib.check(lib.rml_bn_lrelu_pad_backward(
                    hctx, _lib.ptr(x), _lib.ptr(dy), 1 if x.dtype == torch.bfloat16 else 0, n, h, w, c, pad, pad, _lib.ptr(gamma),
                    _lib.ptr(beta), _lib.ptr(mean), _lib.ptr(rstd), slope, _lib.ptr(ws), _lib.ptr(dx), _lib.ptr(dgamma), _lib.ptr(dbeta),
                    _lib.stream_ptr(dev)), "rml_bn_lrelu_pad_backward")
            dbias = torch.zeros_like(ctx.bias_like) if ctx.bias_like is not None else None
            return dx, dgamma.to(gamma.dtype), dbeta.to(beta.dtype), None, None, None, None, None, None, dbias

    _bn_lrelu_pad_function._cls = BnLReluPad
    return BnLReluPad


def bn_lrelu_pad(x, bn, slope=0.2, pad=0, conv_bias=None):
    """``F.pad(F.leaky_relu(bn(x), slope), (0, pad, 0, pad))`` for a training-mode ``nn.BatchNorm2d`` on a CUDA half
    tensor in channels_last layout, as one fused HIP op (csrc/bnact.hip): two streaming passes forward, two backward,
    written straight into the padded layout the next stride-2 TF-'same' convolution reads.  Falls back to the PyTorch ops
    for anything else (CPU, float32, eval mode, channel counts that are not multiples of 8).  ``conv_bias``: see
    ``BnLReluPad.forward`` -- the bias of the producing convolution, which then must NOT have been added to ``x``."""
    torch = _torch()
    import torch.nn.functional as F
    c = x.shape[1]
    fused = (x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and bn.training and bn.track_running_stats and bn.affine
             and bn.momentum is not None and c % 8 == 0 and 256 % (c // 8) == 0)
    if not fused:
        if conv_bias is not None:
            x = x + conv_bias.to(x.dtype).reshape(1, -1, 1, 1)
        y = F.leaky_relu(bn(x), slope)
        return F.pad(y, (0, pad, 0, pad)) if pad else y
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    y = _bn_lrelu_pad_function().apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, slope, pad,
                                       conv_bias)
    _bn_side_effects(bn, conv_bias)
    return y


def _conv1_bn_lrelu_pad_function():
    """First layer of an SGAN branch as ONE autograd node: 3x3 stride-2 convolution of the (already padded) 1-channel
    image -> training-mode batch norm -> LeakyReLU -> pad, entirely in csrc/bnact.hip.  The convolution output (268 MB at
    batch 256) is never stored: it costs 9 FMAs per element to recompute from the tiny image, so every pass (statistics,
    normalise + activation + pad, backward sums, weight gradient) recomputes it; the gradient of the convolution output is
    not materialised either -- it is multiplied with the nine input taps on the fly and summed into the weight gradient.
    (MIOpen's own backward for this shape converts the 268 MB gradient to float32 and runs an im2col GEMM per image:
    3.4 of the 8.5 ms of an update.)"""
    torch = _torch()
    if getattr(_conv1_bn_lrelu_pad_function, "_cls", None) is not None:
        return _conv1_bn_lrelu_pad_function._cls
    import torch.nn.functional as F
    from . import _lib

    class Conv1BnLReluPad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x_padded, weight, conv_bias, gamma, beta, running_mean, running_var, eps, momentum, slope, pad, dtype):
            lib = _lib.load()
            xh = x_padded.to(dtype).contiguous()                               # (N, 1, 2H+1, 2W+1), half
            c = weight.shape[0]
            # [tap][c], rounded like the autocast operand (transpose + rounding in one copy, widening in a second)
            wk = weight.reshape(c, 9).t().to(dtype, memory_format=torch.contiguous_format).float()
            n = xh.shape[0]
            h, w = (xh.shape[2] - 1) // 2, (xh.shape[3] - 1) // 2
            dev = xh.device
            y = torch.empty((n, c, h + pad, w + pad), dtype=dtype, device=dev, memory_format=torch.channels_last)
            mean = torch.empty((c,), dtype=torch.float32, device=dev)
            rstd = torch.empty((c,), dtype=torch.float32, device=dev)
            istat = torch.empty((54,), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                hctx = _lib.context(dev)
                ws = torch.empty((int(lib.rml_bn_workspace_floats(hctx, c)) + 2 * c,), dtype=torch.float32, device=dev)
                _lib.check(lib.rml_conv1_bn_lrelu_pad_forward(
                    hctx, _lib.ptr(xh), _lib.ptr(wk), 1 if dtype == torch.bfloat16 else 0, n, h, w, c, pad, pad, _lib.ptr(gamma),
                    _lib.ptr(beta), float(eps), float(momentum), float(slope), _lib.ptr(running_mean), _lib.ptr(running_var),
                    _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(istat), _lib.ptr(ws), _lib.ptr(y), _lib.stream_ptr(dev)),
                    "rml_conv1_bn_lrelu_pad_forward")
            ctx.save_for_backward(xh, wk, gamma, beta, mean, rstd, istat)
            ctx.meta = (float(slope), int(pad), weight.shape, weight.dtype, h, w)
            sc = fused_step_scope.active
            ctx.bias_like = conv_bias if (sc is None or sc.bias_grads) else None
            return y

        @staticmethod
        def backward(ctx, dy):
            lib = _lib.load()
            xh, wk, gamma, beta, mean, rstd, istat = ctx.saved_tensors
            slope, pad, wshape, wdtype, h, w = ctx.meta
            n, c = xh.shape[0], wk.shape[1]
            dev = xh.device
            if dy.dtype != xh.dtype or not dy.is_contiguous(memory_format=torch.channels_last):
                dy = dy.to(xh.dtype).contiguous(memory_format=torch.channels_last)
            dw = torch.empty((9, c), dtype=torch.float32, device=dev)
            dgamma = torch.empty((c,), dtype=torch.float32, device=dev)
            dbeta = torch.empty((c,), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                hctx = _lib.context(dev)
                ws = torch.empty((int(lib.rml_bn_workspace_floats(hctx, c)) + 2 * c,), dtype=torch.float32, device=dev)
                _lib.check(lib.rml_conv1_bn_lrelu_pad_backward(
                    hctx, _lib.ptr(xh), _lib.ptr(wk), _lib.ptr(dy), 1 if xh.dtype == torch.bfloat16 else 0, n, h, w, c, pad, pad,
                    _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(istat), slope, _lib.ptr(ws), _lib.ptr(dw),
                    _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.stream_ptr(dev)), "rml_conv1_bn_lrelu_pad_backward")
            dbias = torch.zeros_like(ctx.bias_like) if ctx.bias_like is not None else None
            return (None, dw.t().reshape(wshape).to(wdtype), dbias, dgamma.to(gamma.dtype), dbeta.to(beta.dtype),
                    None, None, None, None, None, None, None)

    _conv1_bn_lrelu_pad_function._cls = Conv1BnLReluPad
    return Conv1BnLReluPad


def conv1_bn_lrelu_pad(x_padded, conv, bn, slope, pad, dtype):
    """See ``Conv1BnLReluPad``: ``conv`` is the inner nn.Conv2d(1, C, 3, stride=2) whose input ``x_padded`` already carries
    the TF-'same' zero row/column (N, 1, 2H+1, 2W+1)."""
    torch = _torch()
    y = _conv1_bn_lrelu_pad_function().apply(x_padded, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                             bn.eps, bn.momentum, slope, pad, dtype)
    _bn_side_effects(bn, conv.bias)
    return y


class DeviceAdam:
    """torch.optim.Adam's update (no weight decay, no amsgrad) for a list of CUDA float32 parameters as ONE pass over all of them
    (csrc/optim.hip, ``rml_adam_step``), together with the loss-scale rule of ``torch.amp.GradScaler`` on the device: three
    launches per update instead of torch's fused Adam (two 70 us launches: 29 workgroups on 256 CUs) + the non-finite scan + the
    scaler's one-element kernels.  ``scale`` (a 1-element CUDA float32 tensor shared by the optimizers of a trainer, or None)
    multiplies the loss before ``backward``; ``step(grads)`` tests the gradients, skips the update and halves the scale on a
    non-finite one, doubles it after ``growth_interval`` clean steps in a row.  Exponential averages and the step count are this
    object's.  ``mirror`` (the torch optimizer whose place this takes; optional): its ``param_groups[0]`` is read at every step, so
    a learning-rate change made on the torch optimizer (a scheduler, a manual edit) takes effect here too (sgan.py:206, 214)."""
    MAX_TABLES = 8          # device tables kept (one per set of gradient tensors: eager training re-allocates its gradients)

    def __init__(self, params, lr, betas, eps, scale=None, scaler_state=None, growth=2.0, backoff=0.5, growth_interval=2000, mirror=None):
        torch = _torch()
        self.params = [p for p in params if p.requires_grad]
        if not self.params or any((not p.is_cuda) or p.dtype != torch.float32 for p in self.params):
            raise ValueError("DeviceAdam: CUDA float32 parameters expected")
        dev = self.params[0].device
        self.device = dev
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.growth, self.backoff, self.growth_interval = float(growth), float(backoff), int(growth_interval)
        self.exp_avg = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in self.params]
        self.step_count = torch.zeros((1,), dtype=torch.float32, device=dev)
        self.scale = scale
        # [non-finite count, growth tracker, skip flag]: shared by the optimizers that share a scale (one growth tracker, as in
        # one GradScaler serving two optimizers)
        self.state = scaler_state if scaler_state is not None else torch.zeros((3,), dtype=torch.int32, device=dev)
        self.inv_scale = torch.ones((1,), dtype=torch.float32, device=dev)
        self._tables = {}
        self.mirror = mirror

    def _table(self, grads):
        """device table for this set of gradient tensors (a HIP-graph head owns its own): built once per set"""
        torch = _torch()
        from . import _lib
        key = tuple(0 if g is None else g.data_ptr() for g in grads)
        hit = self._tables.get(key)
        if hit is not None:
            return hit
        rec = int(_lib.load().rml_adam_entry_bytes())
        if rec != 40:
            raise RuntimeError("rml_adam_entry_bytes() = %d, expected 40" % rec)
        rows, start = [], 0
        for p, g, m, v in zip(self.params, grads, self.exp_avg, self.exp_avg_sq):
            if g is None:                                   # a parameter without gradient is left alone (torch.optim does the same)
                continue
            if g.dtype != torch.float32 or g.stride() != p.stride() or g.shape != p.shape:
                raise ValueError("DeviceAdam: gradient layout differs from its parameter's")
            span = 1 + sum((sz - 1) * st for sz, st in zip(p.shape, p.stride())) if p.numel() else 0
            if span != p.numel():
                raise ValueError("DeviceAdam: dense parameters expected")
            rows.append((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), start))
            start += p.numel()
        n = len(rows)
        rows.append((0, 0, 0, 0, start))
        host = np.array(rows, dtype=np.int64).reshape(-1)            # five 8-byte fields per record
        tab = torch.from_numpy(host).to(self.device)
        hit = (tab, n, start)
        if len(self._tables) >= self.MAX_TABLES:            # oldest out (dicts keep insertion order): the cache cannot grow without bound
            self._tables.pop(next(iter(self._tables)))
        self._tables[key] = hit
        return hit

    def step(self, grads=None):
        torch = _torch()
        from . import _lib
        grads = [p.grad for p in self.params] if grads is None else grads
        tab, n, total = self._table(grads)
        if n == 0:
            return
        if self.mirror is not None:                         # hyper-parameters follow the torch optimizer this one mirrors
            g0 = self.mirror.param_groups[0]
            self.lr, self.eps = float(g0["lr"]), float(g0["eps"])
            self.betas = (float(g0["betas"][0]), float(g0["betas"][1]))
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().rml_adam_step(
                _lib.context(self.device), _lib.ptr(tab), n, total, self.lr, self.betas[0], self.betas[1], self.eps,
                _lib.ptr(self.step_count), _lib.ptr(self.scale), _lib.ptr(self.state), _lib.ptr(self.inv_scale),
                1 if self.scale is not None else 0, self.growth, self.backoff, self.growth_interval, _lib.stream_ptr(self.device)),
                "rml_adam_step")

