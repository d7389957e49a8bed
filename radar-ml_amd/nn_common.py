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


def preprocess_projections(samples, rescale, device=None):
    """dnn.py:200-254 / sgan.py:636-690 on the GPU: scale [0,255] -> [-1,1] ((p-127.5)/127.5), resize every
    projection to ``rescale`` with antialiased bicubic interpolation (the PIL ``Image.BICUBIC`` resize of the
    reference), return three (N,1,H,W) tensors in (xz, yz, xy) order."""
    torch = _torch()
    import torch.nn.functional as F
    dev = device or torch.device("cuda", torch.cuda.current_device())
    outs = []
    for i in range(3):
        p = torch.from_numpy(np.stack([np.asarray(s[i], dtype=np.float32) for s in samples])).to(dev)
        p = (p - 127.5) / 127.5
        p = F.interpolate(p.unsqueeze(1), size=tuple(rescale), mode="bicubic", align_corners=False, antialias=True)
        outs.append(p.contiguous(memory_format=torch.channels_last))
    return outs
