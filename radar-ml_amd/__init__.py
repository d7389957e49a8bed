"""radar-ml hot path on MI355X (gfx950): batched 3-D radar volume -> (xz, yz, xy) projections ->
feature rows -> RBF-SVM / linear decision -> labels, behind the reference's own call surface
(``common.process_samples`` / ``clf.predict`` / ``model.predict``).

Host code is Python and mirrors goruck/radar-ml's interface for this path; the work is done by
hand-written HIP kernels in ``libradarml_hip.so`` (C ABI in ``include/radarml.h``, bound with
ctypes in ``_lib.py``).  There is no CPU fallback.
"""
from . import _lib
from ._lib import RadarMLError
from .common import (ProjMask, ProjZoom, DerivedTarget, RADAR_MAX, RADAR_MIN,
                     cartesian_to_spherical, spherical_to_cartesian, calculate_matrix_indices,
                     process_samples, process_volumes, project, derive_targets, feature_len)
from .svm import GpuSVC, GpuCalibratedClassifier, GpuLinearClassifier, KernelMatrix, from_sklearn
from .predict import classifier, classify_batch, calc_proj_zoom
from .synth import synth_volumes
from .augment import DataGenerator, augment_planes, rotation_params

__all__ = [
    "RadarMLError", "ProjMask", "ProjZoom", "DerivedTarget", "RADAR_MAX", "RADAR_MIN",
    "cartesian_to_spherical", "spherical_to_cartesian", "calculate_matrix_indices",
    "process_samples", "process_volumes", "project", "derive_targets", "feature_len",
    "GpuSVC", "GpuCalibratedClassifier", "GpuLinearClassifier", "KernelMatrix", "from_sklearn",
    "classifier", "classify_batch", "calc_proj_zoom", "synth_volumes",
    "DataGenerator", "augment_planes", "rotation_params",
]
