"""ctypes binding of libradarml_hip.so (the C ABI declared in include/radarml.h).

The product path has no CPU fallback: if the library cannot be loaded, or a call fails, a
``RadarMLError`` is raised.  PyTorch-ROCm is used only as the device allocator / stream
provider; tensors cross the boundary as raw device pointers.
"""
import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
# RML_LIB overrides the library path (A/B runs of kernel variants built with radar-ml_amd/build.py --variant)
LIB_PATH = os.environ.get("RML_LIB") or os.path.join(HERE, "libradarml_hip.so")


class RadarMLError(RuntimeError):
    pass


_lib = None
_lock = threading.RLock()      # re-entrant: check() -> load() may run while context() holds it
_ctx = {}

c_void_p, c_int, c_int64, c_uint32, c_uint64, c_float, c_double = (
    C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_double)

# name -> (restype, argtypes): exactly the declarations of include/radarml.h
SIGNATURES = {
    "rml_version": (C.c_char_p, []),
    "rml_last_error": (C.c_char_p, []),
    "rml_ctx_create": (c_int, [c_int, C.POINTER(c_void_p)]),
    "rml_ctx_destroy": (c_int, [c_void_p]),
    "rml_ctx_device": (c_int, [c_void_p]),
    "rml_ctx_set_option": (c_int, [c_void_p, c_int, c_int]),
    "rml_ctx_get_option": (c_int, [c_void_p, c_int, C.POINTER(c_int)]),
    "rml_ctx_reserve_workspace": (c_int, [c_void_p, c_int64]),
    "rml_ctx_workspace_bytes": (c_int64, [c_void_p]),
    "rml_profile_enable": (c_int, [c_void_p, c_int]),
    "rml_profile_read": (c_int, [c_void_p, C.POINTER(c_int64), C.POINTER(c_double), C.POINTER(c_int64)]),
    "rml_profile_read_gemm": (c_int, [c_void_p, C.POINTER(c_int64), C.POINTER(c_double), C.POINTER(c_double)]),
    "rml_feature_len": (c_int64, [c_int, c_int, c_int, c_uint32]),
    "rml_project": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_float, c_uint32,
                            c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rml_project_slices": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_float, c_uint32,
                                   c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rml_project_planes": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "rml_derive_targets": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rml_derive_slice": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_uint32,
                                 c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rml_derive_slice_supported": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "rml_assemble_features": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float,
                                      c_uint32, c_void_p, c_int64, c_void_p]),
    "rml_zoom_features": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_float,
                                  c_uint32, c_void_p, c_int64, c_void_p]),
    "rml_quantize_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_float, c_void_p, c_int64,
                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    "rml_svm_load": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int,
                             c_double, c_double, c_void_p, c_void_p, C.POINTER(c_void_p)]),
    "rml_svm_free": (c_int, [c_void_p, c_void_p]),
    "rml_svm_is_exact": (c_int, [c_void_p]),
    "rml_svm_num_sv": (c_int64, [c_void_p]),
    "rml_svm_dim": (c_int64, [c_void_p]),
    "rml_svm_decision": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                 c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rml_svm_set_platt": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "rml_svm_pairwise_proba": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "rml_svm_kernel_matrix": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "rml_project_svm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_float,
                                c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rml_derive_project_svm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_float, c_uint32, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rml_linear_load": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, C.POINTER(c_void_p)]),
    "rml_linear_free": (c_int, [c_void_p, c_void_p]),
    "rml_linear_decision": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "rml_resize_bicubic": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_float, c_float,
                                   c_void_p, c_int, c_void_p]),
    "rml_dnn_preprocess_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "rml_dnn_preprocess_rows": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                        c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rml_dnn_preprocess_volumes": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int64,
                                           c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rml_dnn_trunk": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p]),
    "rml_dnn_trunk_kblock": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    "rml_dnn_trunk_x3_supported": (c_int, [c_int, c_int]),
    "rml_dnn_trunk_x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int, c_void_p, c_void_p]),
    "rml_dnn_exact_features_scratch_bytes": (c_int64, [c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_int]),
    "rml_dnn_exact_features": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "rml_dnn_top2_gap": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "rml_dnn_guard_apply": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "rml_dnn_dense_workspace_bytes": (c_int64, [c_void_p, c_int64, c_int64]),
    "rml_dnn_dense_tail": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "rml_dnn_dense_tail_f32_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "rml_dnn_dense_tail_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "rml_bn_workspace_floats": (c_int64, [c_void_p, c_int]),
    "rml_bn_lrelu_pad_forward": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                         c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rml_bn_lrelu_pad_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rml_conv1_bn_lrelu_pad_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                               c_void_p, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p]),
    "rml_conv1_bn_lrelu_pad_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_int,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_void_p]),
    "rml_adam_entry_bytes": (c_int, []),
    "rml_adam_step": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_int, c_float, c_float, c_int, c_void_p]),
    "rml_augment": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rml_code_rmw_default": (c_int, [c_int64, c_int64, c_int, c_int]),
    "rml_probe_stream": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "rml_synth_volumes": (c_int, [c_void_p, c_uint64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p]),
}

MODE_MAX, MODE_SLICE, MODE_SUM, MODE_MAX_NAN = 0, 1, 2, 3
# rml_ctx_set_option ids (include/radarml.h RML_OPT_*)
OPT_PROJECT_SHARE_CU, OPT_WAVEFRAME, OPT_LINPLANE, OPT_STAGE_CODES, OPT_SLICE_WAVE, OPT_DERIVE_FUSED, OPT_CODE_RMW, OPT_GEMM_BIG, OPT_CHUNK, OPT_C1_PK = range(1, 11)
OPTIONS = {"project_share_cu": OPT_PROJECT_SHARE_CU, "waveframe": OPT_WAVEFRAME, "linplane": OPT_LINPLANE, "stage_codes": OPT_STAGE_CODES,
           "slice_wave": OPT_SLICE_WAVE, "derive_fused": OPT_DERIVE_FUSED, "code_rmw": OPT_CODE_RMW, "gemm_big": OPT_GEMM_BIG,
           "chunk": OPT_CHUNK, "c1_pk": OPT_C1_PK}
# A/B runs from a shell (tools/profile_round.sh, tools/kbench.py under rocprofv3): these environment variables are read ONCE, here in
# Python, when a context is created, and applied as options -- the library itself never reads the environment
ENV_OPTIONS = {"RML_WAVE_SHARE": "project_share_cu", "RML_WAVEFRAME": "waveframe", "RML_LINPLANE": "linplane", "RML_STAGE_CODES": "stage_codes",
               "RML_SLICE_WAVE": "slice_wave", "RML_DERIVE_FUSED": "derive_fused", "RML_CODE_RMW": "code_rmw", "RML_GEMM_BIG": "gemm_big",
               "RML_CHUNK": "chunk", "RML_C1_PK": "c1_pk"}
AUG_ROTATE, AUG_ZOOM, AUG_NOISE = 0, 1, 2
VOL_F32, VOL_U8 = 0, 1
MODES = {"max": MODE_MAX, "slice": MODE_SLICE, "sum": MODE_SUM, "max_nan": MODE_MAX_NAN}
KERNEL_RBF, KERNEL_LINEAR = 0, 1
PATH_AUTO, PATH_F32, PATH_I8, PATH_F64, PATH_DIGITS = 0, 1, 2, 3, 4
PATHS = {"auto": PATH_AUTO, "f32": PATH_F32, "i8": PATH_I8, "f64": PATH_F64, "digits": PATH_DIGITS}


def load():
    """Load the shared library (building nothing: run ``__graft_entry__.build()`` or
    ``python radar-ml_amd/build.py`` first).  Raises RadarMLError when it is missing."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RadarMLError(
                "libradarml_hip.so is not built (%s missing): run `python radar-ml_amd/build.py`; "
                "there is no CPU fallback for the HIP path" % LIB_PATH)
        # import torch first so that the process-wide HIP runtime (SONAME libamdhip64.so.*) is the one
        # torch allocates with; our library then binds to the same runtime and can use torch's pointers.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is part of the image
            pass
        try:
            lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        except OSError as e:
            raise RadarMLError("cannot load %s: %s" % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                raise RadarMLError("libradarml_hip.so does not export %s" % name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().rml_last_error()
        raise RadarMLError("%s failed (status %d): %s" % (what or "radarml call", rc, msg.decode() if msg else ""))


def context(device=None):
    """One rml_ctx per device, created on first use."""
    import torch
    if not torch.cuda.is_available():
        raise RadarMLError("no HIP device is visible: the radar-ml HIP path needs an MI355X (no CPU fallback)")
    if device is None:
        device = torch.cuda.current_device()
    elif isinstance(device, torch.device):
        device = device.index if device.index is not None else torch.cuda.current_device()
    index = int(device)
    lib = load()
    with _lock:
        h = _ctx.get(device)
    if h is None:
        # create outside the lock (a failing rml_ctx_create must raise, not dead-lock: check() calls load())
        new = c_void_p()
        check(lib.rml_ctx_create(index, C.byref(new)), "rml_ctx_create")
        for var, name in ENV_OPTIONS.items():
            val = os.environ.get(var)
            if val not in (None, ""):
                check(lib.rml_ctx_set_option(new, OPTIONS[name], int(val)), "rml_ctx_set_option(%s from $%s)" % (name, var))
        with _lock:
            h = _ctx.setdefault(device, new)
        if h is not new:                    # another thread won the race
            lib.rml_ctx_destroy(new)
    return h


def set_option(name, value, device=None):
    """rml_ctx_set_option on the device's context; returns the previous value.  ``name``: a key of OPTIONS."""
    lib, ctx = load(), context(device)
    old = C.c_int()
    check(lib.rml_ctx_get_option(ctx, OPTIONS[name], C.byref(old)), "rml_ctx_get_option")
    check(lib.rml_ctx_set_option(ctx, OPTIONS[name], int(value)), "rml_ctx_set_option")
    return old.value


def get_option(name, device=None):
    v = C.c_int()
    check(load().rml_ctx_get_option(context(device), OPTIONS[name], C.byref(v)), "rml_ctx_get_option")
    return v.value


class options:
    """``with _lib.options(gemm_big=0, chunk=4096): ...`` -- context options set for the block and restored after it."""

    def __init__(self, device=None, **kv):
        self.device, self.kv, self.old = device, kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = set_option(k, v, self.device)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v, self.device)
        return False


def device_of(device=None):
    """torch.device of a 'cuda:N' / int / None (current) argument."""
    import torch
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    d = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
    if d.type != "cuda":
        raise RadarMLError("the radar-ml HIP path runs on a HIP device, not %r" % (device,))
    return d if d.index is not None else torch.device("cuda", torch.cuda.current_device())


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def stream_ptr(device=None):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)
