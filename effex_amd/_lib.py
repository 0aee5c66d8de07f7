"""ctypes binding of libfxcorr.so (the C ABI in include/fxcorr.h).

There is no fallback: if the shared library is missing or a symbol cannot be bound, ``load()``
raises.  The library is built in-tree by ``__graft_entry__.build()`` /
``python -m effex_amd.build``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
IN_TREE_LIB = os.path.join(_HERE, "csrc", "libfxcorr.so")
# FXCORR_LIB: developer override to A/B kernel variants built elsewhere (tools/kbench.py); the default is the in-tree
# build, and bench.py / the tests refuse anything else (``is_in_tree()``)
LIB_PATH = os.environ.get("FXCORR_LIB") or IN_TREE_LIB
# the developer build (-DFXC_DEV_KERNELS=1): the shipped kernels plus the reference / A-B kernels the tests and tools/soak.py
# compare them with -- never what a caller gets by default
DEV_LIB = os.path.join(_HERE, "csrc", "libfxcorr_dev.so")


def is_in_tree():
    return os.path.realpath(LIB_PATH) == os.path.realpath(IN_TREE_LIB)


FXC_OK = 0
FXC_ERR_ARG = -1
FXC_ERR_UNSUPPORTED = -2
FXC_ERR_HIP = -3
FXC_ERR_NOMEM = -4
FXC_ERR_NODEVICE = -5
FXC_ERR_STATE = -6
FXC_ERR_COMM = -7
FXC_COMM_ID_BYTES = 128

FXC_MEM_HOST = 0
FXC_MEM_DEVICE = 1
FXC_MEM_DEVICE_TO_PINNED = 2
FXC_MODE_SPECTRUM = 0
FXC_MODE_CONTINUUM = 1
FXC_IQ_C64 = 0
FXC_IQ_U8 = 1
FXC_IQ_C128 = 2
FXC_PATH_GENERIC = 0
FXC_PATH_FUSED = 1
FXC_PATH_STREAM = 2
FXC_PATH_TILED = 3


class FxcInfo(ctypes.Structure):
    _fields_ = [("n_ant", ctypes.c_int32), ("n_baselines", ctypes.c_int32), ("nchan", ctypes.c_int32),
                ("ntaps", ctypes.c_int32), ("num_samp", ctypes.c_int64), ("n_pts", ctypes.c_int64),
                ("path", ctypes.c_int32), ("grid", ctypes.c_int32), ("block", ctypes.c_int32),
                ("lds_bytes", ctypes.c_int32), ("device", ctypes.c_int32), ("cu_count", ctypes.c_int32),
                ("workspace_bytes", ctypes.c_int64), ("specialised", ctypes.c_int32), ("spec_vgprs", ctypes.c_int32),
                ("spec_source", ctypes.c_int32), ("spec_seconds", ctypes.c_float)]


class FxcCommDesc(ctypes.Structure):
    _fields_ = [("ranks_seen", ctypes.c_int32), ("rank_seen", ctypes.c_int32), ("device_seen", ctypes.c_int32),
                ("world_given", ctypes.c_int32), ("rank_given", ctypes.c_int32), ("device_given", ctypes.c_int32),
                ("rccl_version", ctypes.c_int32), ("async_error", ctypes.c_int32), ("reduces", ctypes.c_int64)]


_c = ctypes
_vp = ctypes.c_void_p
# name -> (restype, argtypes); every symbol include/fxcorr.h declares
SIGNATURES = {
    "fxc_version": (_c.c_int, []),
    "fxc_dev_kernels": (_c.c_int, []),
    "fxc_device_count": (_c.c_int, [_c.POINTER(_c.c_int)]),
    "fxc_status_string": (_c.c_char_p, [_c.c_int]),
    "fxc_plan_create": (_c.c_int, [_c.POINTER(_vp), _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int64, _vp, _vp,
                                   _c.c_int]),
    "fxc_plan_destroy": (_c.c_int, [_vp]),
    "fxc_set_stream": (_c.c_int, [_vp, _vp]),
    "fxc_plan_get_info": (_c.c_int, [_vp, _c.POINTER(FxcInfo)]),
    "fxc_spec_probe": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_char_p, _c.c_char_p, _c.c_int]),
    "fxc_last_error": (_c.c_char_p, [_vp]),
    "fxc_set_rot": (_c.c_int, [_vp, _vp]),
    "fxc_channelize": (_c.c_int, [_vp, _vp, _vp, _c.c_int64, _c.c_int]),
    "fxc_fx_accumulate": (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int]),
    "fxc_fx_rows": (_c.c_int, [_vp, _vp, _vp, _c.c_int64, _c.c_int, _c.c_int, _c.c_double]),
    "fxc_acc_reset": (_c.c_int, [_vp]),
    "fxc_acc_export": (_c.c_int, [_vp, _vp]),
    "fxc_finalize_sums": (_c.c_int, [_vp, _vp, _vp, _c.c_int, _c.c_double]),
    "fxc_finalize": (_c.c_int, [_vp, _vp, _c.c_int, _c.c_double, _c.c_int]),
    "fxc_finalize_async": (_c.c_int, [_vp, _c.c_int, _c.c_double, _c.c_int]),
    "fxc_finalize_async_to": (_c.c_int, [_vp, _vp, _c.c_int, _c.c_double, _c.c_int]),
    "fxc_finalize_sums_async": (_c.c_int, [_vp, _vp, _c.c_int, _c.c_double]),
    "fxc_finalize_wait": (_c.c_int, [_vp, _vp]),
    "fxc_finalize_pending": (_c.c_int, [_vp]),
    "fxc_comm_unique_id": (_c.c_int, [_vp]),
    "fxc_comm_create": (_c.c_int, [_c.POINTER(_vp), _c.c_int, _c.c_int, _c.c_int, _vp]),
    "fxc_comm_destroy": (_c.c_int, [_vp]),
    "fxc_reduce": (_c.c_int, [_vp, _vp, _c.c_int]),
    "fxc_comm_info": (_c.c_int, [_vp, _c.POINTER(FxcCommDesc)]),
    "fxc_comm_probe": (_c.c_int, [_vp, _c.POINTER(_c.c_int64)]),
    "fxc_rccl_version": (_c.c_int, [_c.POINTER(_c.c_int), _c.c_char_p, _c.c_int]),
    "fxc_sync": (_c.c_int, [_vp]),
    "fxc_remove_dc": (_c.c_int, [_vp, _vp, _vp, _c.c_int64]),
    "fxc_convert_u8": (_c.c_int, [_vp, _vp, _vp, _c.c_int64, _c.c_int]),
    "fxc_fx_rows_u8": (_c.c_int, [_vp, _vp, _vp, _c.c_int64, _c.c_int, _c.c_int, _c.c_double, _c.c_int]),
    "fxc_fx_accumulate_u8": (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _c.c_int]),
    "fxc_fx_rows_iq": (_c.c_int, [_vp, _vp, _vp, _c.c_int64, _c.c_int, _c.c_int, _c.c_double, _c.c_int, _c.c_int]),
    "fxc_fx_accumulate_iq": (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _c.c_int, _c.c_int]),
    "fxc_host_alloc": (_c.c_int, [_c.POINTER(_vp), _c.c_int64]),
    "fxc_host_free": (_c.c_int, [_vp]),
    "fxc_estimate_delay": (_c.c_int, [_vp, _vp, _vp, _c.c_int64, _c.c_int, _c.c_double, _c.POINTER(_c.c_double)]),
    "fxc_pipe_create": (_c.c_int, [_c.POINTER(_vp), _vp, _c.c_int64, _c.c_int, _c.c_int, _c.c_double]),
    "fxc_pipe_create_u8": (_c.c_int, [_c.POINTER(_vp), _vp, _c.c_int64, _c.c_int, _c.c_int, _c.c_double, _c.c_int]),
    "fxc_pipe_create_iq": (_c.c_int, [_c.POINTER(_vp), _vp, _c.c_int64, _c.c_int, _c.c_int, _c.c_double, _c.c_int,
                                      _c.c_int]),
    "fxc_pipe_acquire": (_c.c_int, [_vp, _c.POINTER(_vp)]),
    "fxc_pipe_submit": (_c.c_int, [_vp]),
    "fxc_pipe_push": (_c.c_int, [_vp, _vp]),
    "fxc_pipe_pop": (_c.c_int, [_vp, _vp]),
    "fxc_pipe_in_flight": (_c.c_int, [_vp]),
    "fxc_pipe_destroy": (_c.c_int, [_vp]),
    "fxc_timer_start": (_c.c_int, [_vp]),
    "fxc_timer_stop": (_c.c_int, [_vp, _c.POINTER(_c.c_double)]),
    "fxc_kernel_profiling": (_c.c_int, [_vp, _c.c_int]),
    "fxc_kernel_time": (_c.c_int, [_vp, _c.POINTER(_c.c_double), _c.POINTER(_c.c_int64), _c.c_int]),
    "fxc_synth_fill": (_c.c_int, [_c.c_int, _vp, _vp, _c.c_uint64, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int64,
                                  _vp, _vp, _c.c_int]),
}

_lib = None
_dev_lib = None


class FxcError(RuntimeError):
    """A libfxcorr call failed; ``status`` is the negative fxc_status."""

    def __init__(self, status, message):
        super().__init__("libfxcorr: {} (status {})".format(message, status))
        self.status = status
        self.message = message


def _bind(path, mode=ctypes.RTLD_GLOBAL):
    lib = ctypes.CDLL(path, mode=mode)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def load(dev=False):
    """Load libfxcorr.so and bind every declared symbol.  Raises if anything is missing.  ``dev``: the developer build
    (tests and tools only)."""
    global _lib, _dev_lib
    if dev:
        if _dev_lib is None:
            if not os.path.isfile(DEV_LIB):
                raise ImportError("libfxcorr_dev.so not found at {} -- `python -m effex_amd.build --dev` builds it".format(DEV_LIB))
            load()                       # (the shipped library first: one HIP runtime in the process)
            _dev_lib = _bind(DEV_LIB, ctypes.RTLD_LOCAL)
            assert _dev_lib.fxc_dev_kernels() == 1
        return _dev_lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            "libfxcorr.so not found at {} — build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
            "effex_amd has no CPU fallback".format(LIB_PATH))
    # torch bundles its own libamdhip64 (same SONAME): load it first so the process has ONE HIP runtime
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = _bind(LIB_PATH)
    _lib = lib
    return lib


def check(status, plan_handle=None, lib=None):
    """Map a non-zero status to the reference's exception types (SURVEY.md §8b 'Errors')."""
    if status == FXC_OK:
        return
    lib = lib or load()
    msg = lib.fxc_last_error(plan_handle)
    msg = msg.decode("utf-8", "replace") if msg else lib.fxc_status_string(status).decode()
    if status == FXC_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)          # cusignal raises NotImplementedError for ntaps > 32
    if status == FXC_ERR_ARG:
        raise ValueError(msg)
    if status == FXC_ERR_NOMEM:
        raise MemoryError(msg)
    raise FxcError(status, msg)
