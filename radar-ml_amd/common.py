"""Host-side mirror of the reference's ``common.py`` for the hot path.

Same names, argument meaning and defaults as goruck/radar-ml ``common.py`` (file:line cited
per symbol); the array work runs in the HIP kernels of ``libradarml_hip.so``.  The geometry
helpers are scalar float64 host code exactly as in the reference (they are not a hot loop).
"""
import collections

import numpy as np

from . import _lib

# Radar scan arena in spherical coordinates -- common.py:25-27
R_MIN, R_MAX, R_RES = 10, 360, 2
THETA_MIN, THETA_MAX, THETA_RES = -42, 42, 4
PHI_MIN, PHI_MAX, PHI_RES = -30, 30, 2
# common.py:30-31
RADAR_MIN = 0.
RADAR_MAX = 255.

# common.py:40 / common.py:43 -- positional order (xz, yz, xy)
ProjMask = collections.namedtuple('ProjMask', ['xz', 'yz', 'xy'])
ProjZoom = collections.namedtuple('ProjZoom', ['xz', 'yz', 'xy'])


def _mask_bits(proj_mask):
    bits = 0
    for i in range(3):
        if proj_mask[i]:
            bits |= 1 << i
    if bits == 0:
        raise ValueError("proj_mask selects no projection")
    return bits


def feature_len(size_x, size_y, size_z, proj_mask=ProjMask(True, True, True)):
    """Length of a feature row: X*Z + Y*Z + X*Y over the selected planes (10010 at (22,31,176))."""
    return int(_lib.load().rml_feature_len(size_x, size_y, size_z, _mask_bits(proj_mask)))


# --------------------------------------------------------------------------------------------
# geometry: common.py:93-121 (scalar host code, float64 like the reference)
# --------------------------------------------------------------------------------------------
def cartesian_to_spherical(x, y, z):
    """common.py:93-97."""
    r = np.sqrt(np.power(x, 2) + np.power(y, 2) + np.power(z, 2))
    phi = np.arctan2(y, z)
    theta = np.arcsin(x / r)
    return (r, np.rad2deg(theta), np.rad2deg(phi))


def spherical_to_cartesian(r, theta, phi):
    """common.py:99-104."""
    theta_rad, phi_rad = np.deg2rad(theta), np.deg2rad(phi)
    x = r * np.sin(theta_rad)
    y = r * np.cos(theta_rad) * np.sin(phi_rad)
    z = r * np.cos(theta_rad) * np.cos(phi_rad)
    return (x, y, z)


def calculate_matrix_indices(x, y, z, size_x, size_y, size_z):
    """common.py:106-121.  Scalars give a tuple of Python ints (truncation toward zero, no
    clamping, exactly like ``int()``); arrays give an (n,3) int32 array (batched form used
    to feed ``project(mode='slice')``)."""
    r, theta, phi = cartesian_to_spherical(x, y, z)
    fi = (theta - THETA_MIN) * (size_x - 1) / (THETA_MAX - THETA_MIN)
    fj = (phi - PHI_MIN) * (size_y - 1) / (PHI_MAX - PHI_MIN)
    fk = (r - R_MIN) * (size_z - 1) / (R_MAX - R_MIN)
    if np.ndim(fi) == 0:
        return (int(fi), int(fj), int(fk))
    return np.stack([np.trunc(fi), np.trunc(fj), np.trunc(fk)], axis=-1).astype(np.int32)


# --------------------------------------------------------------------------------------------
# device helpers
# --------------------------------------------------------------------------------------------
def _torch():
    import torch
    return torch


def _as_device_f32(a, device=None):
    """numpy / torch (any device, any real dtype) -> contiguous float32 CUDA tensor."""
    torch = _torch()
    if not torch.cuda.is_available():
        raise _lib.RadarMLError("no HIP device is visible: the radar-ml HIP path needs an MI355X (no CPU fallback)")
    if isinstance(a, torch.Tensor):
        t = a
    else:
        t = torch.from_numpy(np.ascontiguousarray(a))
    if device is None:
        device = t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())
    return t.to(device=device, dtype=torch.float32, non_blocking=True).contiguous()


def _as_device_volumes(a, device=None):
    """Volumes for the projection entry points: uint8 stays uint8 (the radar's native magnitudes, a quarter of the
    HBM bytes, same results), everything else becomes float32.  Returns (contiguous CUDA tensor, RML_VOL_* code)."""
    torch = _torch()
    is_u8 = (a.dtype == torch.uint8) if isinstance(a, torch.Tensor) else (np.asarray(a).dtype == np.uint8)
    if not is_u8:
        return _as_device_f32(a, device), _lib.VOL_F32
    if not torch.cuda.is_available():
        raise _lib.RadarMLError("no HIP device is visible: the radar-ml HIP path needs an MI355X (no CPU fallback)")
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    if device is None:
        device = t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())
    return t.to(device=device, non_blocking=True).contiguous(), _lib.VOL_U8


def _slice_indices(ijk, B, X, Y, Z, dev, validate=True):
    """ijk as (B,3) or (B,T,3) (numpy / torch, any integer dtype) -> (contiguous int32 CUDA tensor (B*T,3), T).
    Raises like the reference's NumPy indexing would: one (i,j,k) (or T of them) per frame, every index within
    [-size, size) (Python negative-index wrap) -- the kernel never sees an index it would have to clamp.  The range check
    runs on the caller's integers widened to int64, BEFORE the int32 cast (an int64 index must not wrap into range, a narrow
    dtype must not wrap the bound); host arrays are checked
    on the host, device tensors cost one synchronising read-back.  ``validate=False``: indices this package derived itself
    (derive_targets) -- in range by construction, no host synchronisation on the asynchronous path."""
    torch = _torch()
    t = torch.as_tensor(np.asarray(ijk) if not isinstance(ijk, torch.Tensor) else ijk)
    if t.ndim == 1 and B == 1:
        t = t.reshape(1, 3)
    if t.ndim not in (2, 3) or t.shape[-1] != 3:
        raise ValueError("ijk must be (B,3) or (B,T,3)")
    if t.shape[0] != B:
        raise ValueError("ijk must have one (i,j,k) row (or T rows) per frame: got %d for %d frames" % (t.shape[0], B))
    T = 1 if t.ndim == 2 else int(t.shape[1])
    if T < 1:
        raise ValueError("ijk holds no target")
    if t.dtype.is_floating_point or t.dtype == torch.bool:
        raise IndexError("ijk must hold integers")
    if t.dtype in (getattr(torch, "uint16", None), getattr(torch, "uint32", None), getattr(torch, "uint64", None)):
        # torch has no comparison kernels for the wide unsigned types: through NumPy (a uint64 above int64's range is out of
        # bounds for any volume and must not wrap)
        h = t.cpu().numpy()
        if h.size and int(h.max()) > np.iinfo(np.int64).max:
            raise IndexError("index out of bounds: %d" % int(h.max()))
        t = torch.from_numpy(h.astype(np.int64))
    if validate:
        # widened to int64 BEFORE comparing: in a narrow caller dtype -size wraps (uint8) or does not fit (int8 with Z = 128), and
        # an int64 index must be checked before the int32 cast below could wrap it into range
        flat = t.reshape(-1, 3).to(torch.int64)
        size = torch.tensor([X, Y, Z], dtype=torch.int64, device=flat.device)
        bad = ((flat >= size) | (flat < -size)).any(dim=0).cpu().numpy()       # where the tensor lives
        for ax in range(3):
            if bad[ax]:
                raise IndexError("index out of bounds for axis %d with size %d" % (ax, (X, Y, Z)[ax]))
    t = t.to(device=dev, dtype=torch.int32).reshape(-1, 3).contiguous()
    return t, T


def project(volumes, mode="max", ijk=None, return_numpy=None):
    """Batched 3-D -> 2-D projections: returns the reference's tuple ``(xz, yz, xy)``.

    mode='slice': ``yz=V[i,:,:]``, ``xz=V[:,j,:]``, ``xy=V[:,:,k]`` per frame at ``ijk[b]``
    (predict.py:102-107, ground_truth_samples.py:413-419; negative indices wrap like
    Python's).  mode='max': the max-projection named by BASELINE.json (a NaN is ignored, np.fmax);
    mode='max_nan': the same with NumPy's NaN policy (np.max propagates; opt-in, general kernel).  mode='sum': the
    reductions of common.py:51-53.  ``volumes`` is (B,X,Y,Z) or (X,Y,Z); numpy in -> numpy
    out, torch CUDA in -> torch CUDA out.
    """
    torch = _torch()
    lib = _lib.load()
    is_np = not isinstance(volumes, torch.Tensor)
    if return_numpy is None:
        return_numpy = is_np
    single = (volumes.ndim == 3)
    v, vdt = _as_device_volumes(volumes)
    if single:
        v = v.unsqueeze(0)
    if v.ndim != 4:
        raise ValueError("volumes must be (B,X,Y,Z) or (X,Y,Z)")
    B, X, Y, Z = v.shape
    dev = v.device
    ctx = _lib.context(dev)
    m = _lib.MODES[mode]
    ijk_t = None
    if m == _lib.MODE_SLICE:
        if ijk is None:
            raise ValueError("mode='slice' needs ijk")
        ijk_t, T = _slice_indices(ijk, B, X, Y, Z, dev)
        if T != 1:
            raise ValueError("project(): one (i,j,k) per frame; use process_volumes for several targets per frame")
    xz = torch.empty((B, X, Z), dtype=torch.float32, device=dev)
    yz = torch.empty((B, Y, Z), dtype=torch.float32, device=dev)
    xy = torch.empty((B, X, Y), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rml_project_planes(ctx, _lib.ptr(v), vdt, B, X, Y, Z, m, _lib.ptr(ijk_t), _lib.ptr(xz), _lib.ptr(yz),
                                          _lib.ptr(xy), _lib.stream_ptr(dev)), "rml_project_planes")
    out = (xz, yz, xy)
    if single:
        out = tuple(o[0] for o in out)
    if return_numpy:
        out = tuple(o.cpu().numpy() for o in out)
    return out


def process_volumes(volumes, mode="max", ijk=None, proj_mask=ProjMask(xz=True, yz=True, xy=True), scale=False,
                    out=None, codes=False, num_targets=1, device=None, return_ijk=False):
    """Fused batched front door: (B,X,Y,Z) volumes -> (B,D) float32 feature rows in one pass over
    the volumes (projection + ``process_samples`` at zoom 1).  Returns a CUDA tensor; with
    ``codes=True`` returns ``(feat, codes_u8, row_isum, row_isq, row_flags)`` for the exact SVM path; ``codes='only'`` writes no float
    rows at all (``feat`` is None).

    mode='slice' takes ``ijk`` as (B,3) or -- several targets per frame, the reference's ``for target in targets`` over one
    image (predict.py:93-119) -- (B,T,3): the result then has B*T rows, frame-major (row b*T+t), and no volume is
    duplicated.  Without ``ijk`` the ``num_targets`` strongest derived targets of every frame are used (common.py:49-80);
    ``return_ijk=True`` appends the (B,T,3) int32 indices the rows were sliced at to the result.
    """
    torch = _torch()
    lib = _lib.load()
    v, vdt = _as_device_volumes(volumes, device)
    if v.ndim == 3:
        v = v.unsqueeze(0)
    B, X, Y, Z = v.shape
    dev = v.device
    ctx = _lib.context(dev)
    bits = _mask_bits(proj_mask)
    D = int(lib.rml_feature_len(X, Y, Z, bits))
    m = _lib.MODES[mode]
    ijk_t = None
    T = 1
    fused_derive = False
    if m == _lib.MODE_SLICE:
        derived = False
        if ijk is None:
            # no SDK targets: derive the strongest return(s) per frame on the GPU (common.py:49-80) and slice there -- one pass
            # over the volumes where the shape has the fused kernel (rml_derive_slice), two launches otherwise
            derived = True
            T = int(num_targets)
            fused_derive = bool(lib.rml_derive_slice_supported(_lib.context(v.device), _lib.ptr(v), vdt, X, Y, Z, T))
            if not fused_derive:
                ijk = derive_targets(v, num_targets)
        if not fused_derive:
            ijk_t, T = _slice_indices(ijk, B, X, Y, Z, dev, validate=not derived)
    R = B * T                           # output rows
    feat = out if (out is not None or codes == "only") else torch.empty((R, D), dtype=torch.float32, device=dev)
    if feat is not None and (feat.shape[0] != R or feat.shape[1] < D or feat.device != dev):
        raise ValueError("out must be a (%d, >=%d) float32 tensor on %s" % (R, D, dev))
    q = isum = isq = flags = None
    ldq = 0
    if codes == "only":
        # the projections as uint8 codes and nothing else (+ their statistics and the per-row "all integers in [0,255]" flags):
        # the operand of the exact SVM path without the float rows through HBM
        if out is not None:
            raise ValueError("codes='only' writes no float rows")
        feat = None
    if codes:
        ldq = (D + 127) // 128 * 128
        q = torch.empty((R, ldq), dtype=torch.uint8, device=dev)
        isum = torch.empty((R,), dtype=torch.int32, device=dev)
        isq = torch.empty((R,), dtype=torch.int64, device=dev)
        flags = torch.empty((R,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        if fused_derive:
            ijk_t = torch.empty((B, T, 3), dtype=torch.int32, device=dev) if return_ijk else None
            _lib.check(lib.rml_derive_slice(ctx, _lib.ptr(v), vdt, B, X, Y, Z, T, _lib.ptr(ijk_t), None, float(RADAR_MAX) if scale else 0.0,
                                            bits, _lib.ptr(feat), feat.stride(0) if feat is not None else 0, _lib.ptr(q), ldq, _lib.ptr(isum), _lib.ptr(isq),
                                            _lib.ptr(flags), _lib.stream_ptr(dev)), "rml_derive_slice")
        elif T > 1:
            _lib.check(lib.rml_project_slices(ctx, _lib.ptr(v), vdt, B, X, Y, Z, T, _lib.ptr(ijk_t), float(RADAR_MAX) if scale else 0.0,
                                              bits, _lib.ptr(feat), feat.stride(0) if feat is not None else 0, _lib.ptr(q), ldq, _lib.ptr(isum), _lib.ptr(isq),
                                              _lib.ptr(flags), _lib.stream_ptr(dev)), "rml_project_slices")
        else:
            _lib.check(lib.rml_project(ctx, _lib.ptr(v), vdt, B, X, Y, Z, m, _lib.ptr(ijk_t), float(RADAR_MAX) if scale else 0.0,
                                       bits, _lib.ptr(feat), feat.stride(0) if feat is not None else 0, _lib.ptr(q), ldq, _lib.ptr(isum), _lib.ptr(isq),
                                       _lib.ptr(flags), _lib.stream_ptr(dev)), "rml_project")
    res = (feat, q, isum, isq, flags) if codes else feat
    if return_ijk:
        ijk_r = ijk_t.reshape(B, T, 3) if ijk_t is not None else None
        return (res + (ijk_r,)) if codes else (res, ijk_r)
    return res


def process_samples(samples, proj_mask=ProjMask(xz=True, yz=True, xy=True),
                    proj_zoom=ProjZoom(xz=[1.0, 1.0], yz=[1.0, 1.0], xy=[1.0, 1.0]), scale=False):
    """Prepare samples for training or predictions -- drop-in for common.py:123-149.

    Args:
        samples (list of tuples of np arrays): radar projections [(xz, yz, xy)].
        proj_mask (tuple of bool): projection(s) to use (xz, yz, xy) -- indexed positionally.
        proj_zoom (tuple of list of floats): projection zoom factors (xz, yz, xy).
        scale (bool): if True scales each feature to [0, 1] (``/ RADAR_MAX``).

    Returns:
        np.ndarray (N, D) float32, rows = np.concatenate((xz, yz, xy) selected, axis=None).

    Zoom 1.0 (the value calc_proj_zoom produces whenever the predict arena equals the train
    arena, predict.log:21) is handled as the identity: SciPy's order-3 spline round trip at
    zoom 1 differs from it by <= 1.3e-13 absolute on 0..255 data.  Any other
    zoom runs SciPy's algorithm (order-3 spline, mode 'constant', prefilter) in ``k_zoom``.
    """
    torch = _torch()
    lib = _lib.load()
    unit = all((not proj_mask[i]) or all(abs(float(z) - 1.0) <= 1e-12 for z in np.ravel(proj_zoom[i])) for i in range(3))
    samples = list(samples)
    n = len(samples)
    if n == 0:
        return np.array([])          # np.array([]) is what the reference returns for no samples
    # the selected planes of all samples in ONE host buffer, plane-major ([all xz | all yz | all xy]): one upload instead of three
    # (a per-target call of predict.py:112-116 is dominated by its host-device hops; the row-major concatenation, the scaling and a
    # non-unit zoom are the device's: rml_assemble_features / rml_zoom_features)
    shapes = []
    for i in range(3):
        if not proj_mask[i]:
            shapes.append(None)
            continue
        shp = np.shape(samples[0][i])
        if len(shp) != 2:
            raise ValueError("projection %d of sample 0 is not 2-D" % i)
        for s in samples:
            if np.shape(s[i]) != shp:
                # the reference's np.array([...]) raises on ragged rows
                raise ValueError("setting an array element with a sequence: ragged projection shapes")
        shapes.append(tuple(int(v) for v in shp))
    hostbuf = np.empty(sum(n * sh[0] * sh[1] for sh in shapes if sh is not None), dtype=np.float32)
    planes, offs, off = [], [], 0
    for i in range(3):
        if shapes[i] is None:
            planes.append(None); offs.append(None)
            continue
        cnt = n * shapes[i][0] * shapes[i][1]
        view = hostbuf[off:off + cnt].reshape((n,) + shapes[i])
        for si, smp in enumerate(samples):
            view[si] = smp[i]
        planes.append(view); offs.append((off, cnt))
        off += cnt
    # grid sizes from the selected planes: xz (X,Z), yz (Y,Z), xy (X,Y)
    X = planes[0].shape[1] if planes[0] is not None else (planes[2].shape[1] if planes[2] is not None else 1)
    Z = planes[0].shape[2] if planes[0] is not None else (planes[1].shape[2] if planes[1] is not None else 1)
    Y = planes[1].shape[1] if planes[1] is not None else (planes[2].shape[2] if planes[2] is not None else 1)
    expect = [(X, Z), (Y, Z), (X, Y)]
    for i in range(3):
        if planes[i] is not None and tuple(planes[i].shape[1:]) != expect[i]:
            raise ValueError("projection shapes are inconsistent: got %s for plane %d, expected %s"
                             % (planes[i].shape[1:], i, expect[i]))
    bits = _mask_bits(proj_mask)
    D = int(lib.rml_feature_len(X, Y, Z, bits))
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    ctx = _lib.context(dev)
    dbuf = torch.from_numpy(hostbuf).to(dev)
    dplanes = [None if planes[i] is None else dbuf[offs[i][0]:offs[i][0] + offs[i][1]].view(planes[i].shape) for i in range(3)]
    if not unit:
        import ctypes as C
        # output shapes exactly as scipy.ndimage.zoom computes them: int(round(size * zoom)), Python round()
        oshape = []
        for i in range(3):
            zf = list(np.ravel(proj_zoom[i]).astype(float)) if np.ndim(proj_zoom[i]) else [float(proj_zoom[i])] * 2
            oshape += [int(round(expect[i][0] * zf[0])), int(round(expect[i][1] * zf[1]))]
        D = sum(oshape[2 * i] * oshape[2 * i + 1] for i in range(3) if proj_mask[i])
        feat = torch.empty((n, D), dtype=torch.float32, device=dev)
        arr = (C.c_int32 * 6)(*oshape)
        with torch.cuda.device(dev):
            _lib.check(lib.rml_zoom_features(ctx, _lib.ptr(dplanes[0]), _lib.ptr(dplanes[1]), _lib.ptr(dplanes[2]), n, X, Y, Z,
                                             C.cast(arr, C.c_void_p), float(RADAR_MAX) if scale else 0.0, bits,
                                             _lib.ptr(feat), D, _lib.stream_ptr(dev)), "rml_zoom_features")
        return feat.cpu().numpy()
    feat = torch.empty((n, D), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rml_assemble_features(ctx, _lib.ptr(dplanes[0]), _lib.ptr(dplanes[1]), _lib.ptr(dplanes[2]), n, X, Y, Z,
                                             float(RADAR_MAX) if scale else 0.0, bits, _lib.ptr(feat), D,
                                             _lib.stream_ptr(dev)), "rml_assemble_features")
    return feat.cpu().numpy()


def derive_targets(volumes, num_targets=1, return_profiles=False):
    """Batched DerivedTarget.get_derived_targets (common.py:49-80): (B,num_targets,3) int32
    (i,j,k) triples, ascending by energy; optionally the three energy profiles."""
    torch = _torch()
    lib = _lib.load()
    v, vdt = _as_device_volumes(volumes)
    if v.ndim == 3:
        v = v.unsqueeze(0)
    B, X, Y, Z = v.shape
    dev = v.device
    ctx = _lib.context(dev)
    ijk = torch.empty((B, num_targets, 3), dtype=torch.int32, device=dev)
    prof = torch.empty((B, X + Y + Z), dtype=torch.float32, device=dev) if return_profiles else None
    with torch.cuda.device(dev):
        _lib.check(lib.rml_derive_targets(ctx, _lib.ptr(v), vdt, B, X, Y, Z, int(num_targets), _lib.ptr(ijk), _lib.ptr(prof),
                                          _lib.stream_ptr(dev)), "rml_derive_targets")
    if return_profiles:
        return ijk, prof
    return ijk


class DerivedTarget(collections.namedtuple('DerivedTarget',
                                           ['xPosCm', 'yPosCm', 'zPosCm', 'amplitude', 'i', 'j', 'k'])):
    """Radar targets derived from the raw image -- common.py:45-80."""

    @staticmethod
    def get_derived_targets(radar_data, size_x, size_y, size_z, num_targets=1):
        """Same signature and return value as common.py:49; the sum reductions and the top-k run
        on the GPU, the index -> (theta, phi, r) -> (x, y, z) mapping of ``make``
        (common.py:62-79) on the host."""
        ijk = derive_targets(np.asarray(radar_data).reshape(size_x, size_y, size_z), num_targets).cpu().numpy()[0]
        out = []
        for i, j, k in ijk:
            i, j, k = int(i), int(j), int(k)
            theta = THETA_MIN + i * (THETA_MAX - THETA_MIN) / (size_x - 1)
            phi = PHI_MIN + j * (PHI_MAX - PHI_MIN) / (size_y - 1)
            r = R_MIN + k * (R_MAX - R_MIN) / (size_z - 1)
            x, y, z = spherical_to_cartesian(r, theta, phi)
            out.append(DerivedTarget(xPosCm=x, yPosCm=y, zPosCm=z, amplitude=None, i=i, j=j, k=k))
        return out
