"""On-disk data-set format of the reference (datasets/README.md:8-20; written by
ground_truth_samples.py:561-587, read by train.py:640-667): a pickle of
``{'samples': [(xz, yz, xy), ...], 'labels': [str, ...]}`` with float32 projections in [0, RADAR_MAX].

``load_dataset`` turns such files into three stacked arrays (N,X,Z) (N,Y,Z) (N,X,Y) + labels -- the layout the
GPU feature kernels consume directly (``features_from_dataset``), which removes the per-sample Python list
handling that dominates ``common.process_samples`` in the reference (1.3 ms/sample, train_svc.log:18-19).
"""
import pickle

import numpy as np

from .common import ProjMask, RADAR_MAX, _mask_bits
from . import _lib


def load_dataset(paths, label_alias=None, desired_labels=None):
    """Read one or more data-set pickles (train.py:640-663: concatenate, alias, filter by label).
    Returns (xz, yz, xy, labels): float32 arrays stacked over samples and a list of label strings."""
    if isinstance(paths, (str, bytes)):
        paths = [paths]
    samples, labels = [], []
    for p in paths:
        with open(p, "rb") as fp:
            d = pickle.load(fp)
        samples.extend(d["samples"])
        labels.extend(d["labels"])
    if label_alias:
        labels = [label_alias.get(l, l) for l in labels]
    if desired_labels is not None:
        keep = [i for i, l in enumerate(labels) if l in desired_labels]
        samples = [samples[i] for i in keep]
        labels = [labels[i] for i in keep]
    if not samples:
        z = np.zeros((0, 0, 0), np.float32)
        return z, z, z, []
    shapes = [tuple(np.shape(p)) for p in samples[0]]
    for s in samples:
        if [tuple(np.shape(p)) for p in s] != shapes:
            raise ValueError("samples in a data set must share one radar arena (ragged projection shapes)")
    xz = np.stack([np.asarray(s[0], dtype=np.float32) for s in samples])
    yz = np.stack([np.asarray(s[1], dtype=np.float32) for s in samples])
    xy = np.stack([np.asarray(s[2], dtype=np.float32) for s in samples])
    return xz, yz, xy, labels


def save_dataset(path, xz, yz, xy, labels):
    """Write (or extend, like ground_truth_samples.py:561-587) a data-set pickle in the reference's format."""
    samples = [(np.asarray(a, np.float32), np.asarray(b, np.float32), np.asarray(c, np.float32))
               for a, b, c in zip(xz, yz, xy)]
    try:
        with open(path, "rb") as fp:
            d = pickle.load(fp)
        d["samples"].extend(samples)
        d["labels"].extend(list(labels))
    except FileNotFoundError:
        d = {"samples": samples, "labels": list(labels)}
    with open(path, "wb") as fp:
        pickle.dump(d, fp)
    return len(d["labels"])


def features_from_dataset(xz, yz, xy, proj_mask=ProjMask(True, True, True), scale=True, device=None):
    """(N,D) float32 CUDA feature rows from stacked projections: common.process_samples at zoom 1
    (``scale=True`` = the ``p / RADAR_MAX`` of train.py:667) without the Python per-sample loop."""
    import torch
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    ctx = _lib.context(dev)
    n, X, Z = xz.shape
    Y = yz.shape[1]
    bits = _mask_bits(proj_mask)
    D = int(lib.rml_feature_len(X, Y, Z, bits))
    d = [torch.as_tensor(a, dtype=torch.float32).to(dev).contiguous() for a in (xz, yz, xy)]
    feat = torch.empty((n, D), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rml_assemble_features(ctx, _lib.ptr(d[0]), _lib.ptr(d[1]), _lib.ptr(d[2]), n, X, Y, Z,
                                             float(RADAR_MAX) if scale else 0.0, bits, _lib.ptr(feat), D,
                                             _lib.stream_ptr(dev)), "rml_assemble_features")
    return feat
