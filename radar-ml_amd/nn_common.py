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


class SamePadConv2d:
    """Factory: Conv2d with TF 'same' padding resolved at call time (nn.Module defined lazily to keep torch
    imports out of module import)."""


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
