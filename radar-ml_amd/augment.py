"""GPU twin of the reference's ``train.DataGenerator`` (train.py:32-185): rotation, clipped zoom and sparse Gaussian noise of
radar projections with class balancing, for augmenting the training set (train.py:496-517).

Same constructor, same ``flow(x, y, batch_size)`` generator protocol, same output order (per sample and repetition: the
rotated tuple, the zoomed tuple, the noisy tuple) and -- because the random draws are made on the host with the very calls
the reference makes (``np.random.uniform`` for the angles and the zoom factor, ``np.random.Generator(np.random.PCG64()).normal``
for the noise) in the very same order -- the same data set for the same seeds.  The array work (SciPy's order-3 spline
``ndimage.rotate`` / ``ndimage.zoom``, the clamp) runs in ``csrc/augment.hip``.
"""
import collections

import numpy as np

from . import _lib


def _torch():
    import torch
    return torch


def rotation_params(angle, shape):
    """The affine map ``scipy.ndimage.rotate(p, angle, reshape=False)`` hands to ``affine_transform``
    (scipy/ndimage/_interpolation.py): matrix [[c, s], [-s, c]], offset = centre - matrix @ centre."""
    a = np.deg2rad(float(angle))
    c, s = np.cos(a), np.sin(a)
    # exact at the multiples of 90 degrees, like scipy.special.cosdg / sindg
    if float(angle) % 90.0 == 0.0:
        k = int(round(float(angle) / 90.0)) % 4
        c, s = ((1.0, 0.0), (0.0, 1.0), (-1.0, 0.0), (0.0, -1.0))[k]
    m = np.array([[c, s], [-s, c]], dtype=np.float64)
    centre = (np.asarray(shape, dtype=np.float64) - 1.0) / 2.0
    off = centre - m @ centre
    return np.array([m[0, 0], m[0, 1], m[1, 0], m[1, 1], off[0], off[1]], dtype=np.float64)


def augment_planes(planes, op, params, device=None):
    """One batch of equally shaped planes through ``rml_augment``.  planes: (B,H,W) float32 (numpy or CUDA tensor);
    op: 'rotate' (params (B,6), see ``rotation_params``), 'zoom' (params (B,) factors) or 'noise' (params (B,) draws).
    Returns a CUDA float32 tensor (B,H,W)."""
    torch = _torch()
    lib = _lib.load()
    dev = _lib.device_of(device)
    src = planes if isinstance(planes, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(planes, dtype=np.float32))
    src = src.to(device=dev, dtype=torch.float32).contiguous()
    if src.ndim != 3:
        raise ValueError("planes must be (B,H,W)")
    B, H, W = (int(v) for v in src.shape)
    code = {"rotate": _lib.AUG_ROTATE, "zoom": _lib.AUG_ZOOM, "noise": _lib.AUG_NOISE}[op]
    par = torch.from_numpy(np.ascontiguousarray(params, dtype=np.float64).reshape(B, -1)).to(dev)
    if par.shape[1] != (6 if op == "rotate" else 1):
        raise ValueError("params: %s per plane expected" % ("6 values" if op == "rotate" else "1 value"))
    dst = torch.empty_like(src)
    with torch.cuda.device(dev):
        _lib.check(lib.rml_augment(_lib.context(dev), code, _lib.ptr(src), B, H, W, _lib.ptr(par), _lib.ptr(dst),
                                   _lib.stream_ptr(dev)), "rml_augment")
    return dst


class DataGenerator(object):
    """Generate augmented radar data (train.py:32-51): same arguments, same ``flow``."""

    def __init__(self, rotation_range=None, zoom_range=None, noise_sd=None, balance=False, device=None):
        self.rotation_range = rotation_range
        self.zoom_range = zoom_range
        self.noise_sd = noise_sd
        self.balance = balance
        self.device = device

    def _augment(self, x_batch, y_batch, class_weights):
        """train.py:84-185 ``augment``: the draws here, in the reference's order; the arrays on the GPU."""
        rg = np.random.Generator(np.random.PCG64())                     # train.py:85
        jobs = []                   # (kind, sample index, per-projection parameters)
        aug_y = []
        for si, (xb, yb) in enumerate(zip(x_batch, y_batch)):
            for _ in range(int(np.round(class_weights[yb]))):
                if self.rotation_range is not None:
                    ang = [np.random.uniform(-1 * self.rotation_range, self.rotation_range) for _ in xb]     # one per projection
                    jobs.append(("rotate", si, ang)); aug_y.append(yb)
                if self.zoom_range is not None:
                    zf = np.random.uniform(1.0 - self.zoom_range, 1.0 + self.zoom_range)                     # one per tuple
                    jobs.append(("zoom", si, [zf] * len(xb))); aug_y.append(yb)
                if self.noise_sd is not None:
                    nz = [rg.normal(scale=self.noise_sd) for _ in xb]                                         # one per projection
                    jobs.append(("noise", si, nz)); aug_y.append(yb)
        if not jobs:
            return [], np.array(aug_y)
        nproj = len(x_batch[0])
        outs = [[None] * nproj for _ in jobs]
        for pi in range(nproj):
            for kind in ("rotate", "zoom", "noise"):
                idx = [j for j, job in enumerate(jobs) if job[0] == kind]
                if not idx:
                    continue
                planes = np.stack([np.asarray(x_batch[jobs[j][1]][pi], dtype=np.float32) for j in idx])
                if kind == "rotate":
                    par = np.stack([rotation_params(jobs[j][2][pi], planes.shape[1:]) for j in idx])
                else:
                    par = np.array([jobs[j][2][pi] for j in idx], dtype=np.float64)
                res = augment_planes(planes, kind, par, self.device).cpu().numpy()
                for k, j in enumerate(idx):
                    outs[j][pi] = res[k]
        return [tuple(o) for o in outs], np.array(aug_y)

    def flow(self, x, y, batch_size=32, save_to_dir=None, save_prefix='./datasets/augment'):
        """Yield batches of augmented radar data, forever (train.py:52-83, 187-214); the caller breaks the loop."""
        c = collections.Counter(y)
        mc = c.most_common()
        if self.balance:
            class_weights = {k: mc[0][1] / cnt for k, cnt in mc}
        else:
            class_weights = {k: 1 for k, _ in mc}
        batch = 0
        while True:
            for pos in range(0, len(x), batch_size):
                remaining = len(x) - pos
                end = remaining if remaining < batch_size else batch_size
                x_batch = x[pos:pos + end]
                y_batch = y[pos:pos + end]
                yield self._augment(x_batch, y_batch, class_weights)
                if save_to_dir is not None:
                    import os
                    import pickle
                    fname = f'batch_{str(batch)}_{str(pos)}.pickle'
                    with open(os.path.join(save_prefix, fname), 'wb') as fp:
                        pickle.dump({'x_batch': x_batch, 'y_batch': y_batch}, fp)
            batch += 1
