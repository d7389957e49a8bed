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

    # -- the random draws, on the host, in the reference's order ---------------------------------------------
    def _draw_jobs(self, x_batch, y_batch, class_weights):
        """train.py:84-185 ``augment`` as a job list: [(kind, sample index, one parameter per projection)] and the labels.
        Order of the draws = order of the reference's calls: per sample and repetition the angles of the rotated tuple (one
        ``np.random.uniform`` per projection), the factor of the zoomed tuple (one per tuple), the noise of the noisy tuple (one
        ``Generator(PCG64()).normal`` per projection)."""
        rg = np.random.Generator(np.random.PCG64())                     # train.py:85
        jobs, labels = [], []
        for si, (xb, yb) in enumerate(zip(x_batch, y_batch)):
            for _ in range(int(np.round(class_weights[yb]))):
                if self.rotation_range is not None:
                    jobs.append(("rotate", si, [np.random.uniform(-1 * self.rotation_range, self.rotation_range) for _ in xb]))
                    labels.append(yb)
                if self.zoom_range is not None:
                    zf = np.random.uniform(1.0 - self.zoom_range, 1.0 + self.zoom_range)
                    jobs.append(("zoom", si, [zf] * len(xb)))
                    labels.append(yb)
                if self.noise_sd is not None:
                    jobs.append(("noise", si, [rg.normal(scale=self.noise_sd) for _ in xb]))
                    labels.append(yb)
        return jobs, labels

    def _run_jobs(self, x_batch, jobs, to_host):
        """The array work of a job list on the GPU.  Planes are batched per (projection, kind, plane shape): samples of
        different radar arenas may sit in one data set (the reference augments every sample on its own and only
        ``process_samples`` brings them to a common size).  Returns per job a tuple of planes: numpy arrays of the input's
        dtype (``to_host``) or CUDA float32 tensors."""
        nproj = len(x_batch[0]) if len(x_batch) else 0
        outs = [[None] * nproj for _ in jobs]
        for pi in range(nproj):
            groups = {}
            for j, (kind, si, _) in enumerate(jobs):
                groups.setdefault((kind, tuple(np.shape(x_batch[si][pi]))), []).append(j)
            for (kind, shape), idx in groups.items():
                planes = np.stack([np.asarray(x_batch[jobs[j][1]][pi], dtype=np.float32) for j in idx])
                if kind == "rotate":
                    par = np.stack([rotation_params(jobs[j][2][pi], shape) for j in idx])
                else:
                    par = np.array([jobs[j][2][pi] for j in idx], dtype=np.float64)
                res = augment_planes(planes, kind, par, self.device)
                if to_host:
                    res = res.cpu().numpy()
                for k, j in enumerate(idx):
                    plane = res[k]
                    if to_host:
                        plane = plane.astype(np.asarray(x_batch[jobs[j][1]][pi]).dtype, copy=False)
                    outs[j][pi] = plane
        return [tuple(o) for o in outs]

    def _augment(self, x_batch, y_batch, class_weights):
        jobs, labels = self._draw_jobs(x_batch, y_batch, class_weights)
        if not jobs:
            return [], np.array(labels)
        return self._run_jobs(x_batch, jobs, to_host=True), np.array(labels)

    def _class_weights(self, y):
        """Repetitions per class (train.py:187-196): with ``balance`` the most frequent class counts once and the others
        count (its size / their size), rounded when used; without it every class counts once."""
        counts = collections.Counter(y).most_common()
        top = counts[0][1]
        return {cls: (top / n if self.balance else 1) for cls, n in counts}

    def flow(self, x, y, batch_size=32, save_to_dir=None, save_prefix='./datasets/augment'):
        """Generator protocol of train.py:52-83 (same keyword arguments): walks ``x`` / ``y`` in slices of ``batch_size`` (the
        last one shorter), yields ``(augmented tuples, augmented labels)`` per slice and starts over when the data set is
        exhausted -- it never stops, the caller counts the batches (train.py:508-515).  With ``save_to_dir`` set, every INPUT
        slice is pickled after its batch was consumed, as the reference's debugging aid does (train.py:207-211: the file goes
        under ``save_prefix``, named by epoch and slice start; ``save_to_dir`` itself is only the switch there too)."""
        weights = self._class_weights(y)
        n = len(x)
        epoch = 0
        while True:
            for lo in range(0, n, batch_size):
                hi = min(lo + batch_size, n)
                yield self._augment(x[lo:hi], y[lo:hi], weights)
                if save_to_dir is not None:
                    import os
                    import pickle
                    with open(os.path.join(save_prefix, 'batch_%d_%d.pickle' % (epoch, lo)), 'wb') as fp:
                        pickle.dump({'x_batch': x[lo:hi], 'y_batch': y[lo:hi]}, fp)
            epoch += 1

    def augment_dataset(self, x, y):
        """One pass of ``flow`` over the whole data set (what an epoch of train.py:506-515 appends to the training set) as ONE
        batched device job: the draws are made for every sample first, in the order the reference makes them batch by batch
        (its ``np.random.uniform`` stream simply continues across batches; its noise generator is a fresh, unseeded
        ``Generator(PCG64())`` per batch, here one per call), then every (projection, kind, shape) group is one ``rml_augment``
        launch and the results stay on the GPU.  Returns ``(tuples of CUDA float32 planes, labels as numpy)``; feed the planes
        to ``process_samples`` / the SVM front doors without a host round trip."""
        weights = self._class_weights(y)
        jobs, labels = self._draw_jobs(x, y, weights)
        if not jobs:
            return [], np.array(labels)
        return self._run_jobs(x, jobs, to_host=False), np.array(labels)
