"""Host-side mirror of the classification helpers of the reference's ``predict.py``."""
import numpy as np

from .common import ProjZoom


def calc_proj_zoom(train_size_x, train_size_y, train_size_z, size_x, size_y, size_z):
    """Projection zoom factors for a predict arena -- predict.py:34-54 (same packing:
    ProjZoom(xy=[x,y], xz=[x,z], yz=[y,z]))."""
    x_zoom = train_size_x / size_x
    y_zoom = train_size_y / size_y
    z_zoom = train_size_z / size_z
    return ProjZoom(xy=[x_zoom, y_zoom], xz=[x_zoom, z_zoom], yz=[y_zoom, z_zoom])


def classifier(observation, model, le, min_proba=0.7):
    """Classify a single radar image -- predict.py:56-70, same signature and return value
    ``(name, proba)``; ``model`` is any object with ``predict_proba`` (e.g. a
    GpuCalibratedClassifier), ``le`` anything with ``classes_``."""
    preds = model.predict_proba(np.asarray(observation).reshape(1, -1))[0]
    j = np.argmax(preds)
    proba = preds[j]
    name = le.classes_[j] if proba >= min_proba else 'Unknown'
    return name, proba


def classify_batch(proba, class_names, min_proba=0.7):
    """Batched counterpart of :func:`classifier` on an (N,C) probability array: returns
    (names list, max-probability array); rows below ``min_proba`` are 'Unknown'."""
    proba = np.asarray(proba)
    j = np.argmax(proba, axis=1)
    p = proba[np.arange(proba.shape[0]), j]
    names = [class_names[jj] if pp >= min_proba else 'Unknown' for jj, pp in zip(j, p)]
    return names, p
