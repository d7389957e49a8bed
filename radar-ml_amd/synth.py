"""Synthetic radar return volumes generated in HBM (bench / large-scale tests)."""
from . import _lib


def synth_volumes(n_frames, size_x, size_y, size_z, seed=1234, frame0=0, n_classes=3, device=None, out=None):
    """(n_frames, X, Y, Z) float32 CUDA tensor of integer-valued sparse Gaussian-blob returns and the
    (n_frames,) int32 class of every frame.  Frame f of a data set is a pure function of
    (seed, frame0 + f), so ranks generate disjoint slabs of one global data set."""
    import torch
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    ctx = _lib.context(dev)
    v = out if out is not None else torch.empty((n_frames, size_x, size_y, size_z), dtype=torch.float32, device=dev)
    cls = torch.empty((n_frames,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rml_synth_volumes(ctx, int(seed), int(frame0), int(n_frames), size_x, size_y, size_z, n_classes,
                                         _lib.ptr(v), _lib.ptr(cls), _lib.stream_ptr(dev)), "rml_synth_volumes")
    return v, cls
