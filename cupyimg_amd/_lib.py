"""ctypes binding of libmi355img.so (the C-ABI declared in include/mi355img.h).

The library is the product: there is no CPU fallback.  If the shared object is
missing it is built with hipcc on first use; if that is impossible the import
fails loudly.
"""
import ctypes
import os
import threading

from . import _build

MI_MAX_NDIM = 8

MI_OK = 0
MI_ERR_INVALID_ARG = -1
MI_ERR_UNSUPPORTED = -2
MI_ERR_NOMEM = -3
MI_ERR_NOT_CONTIGUOUS = -4
MI_ERR_RCCL = -5
MI_ERR_INTERNAL = -6

MODE_CODES = {
    "reflect": 0, "grid-mirror": 0, "constant": 1, "nearest": 2, "mirror": 3,
    "wrap": 4, "grid-wrap": 5, "grid-constant": 6,
}


class MiArray(ctypes.Structure):
    _fields_ = [
        ("data", ctypes.c_void_p),
        ("dtype", ctypes.c_int32),
        ("ndim", ctypes.c_int32),
        ("shape", ctypes.c_int64 * MI_MAX_NDIM),
        ("strides", ctypes.c_int64 * MI_MAX_NDIM),
    ]


class Unsupported(Exception):
    """The library has no kernel for this (valid) request; callers fall back
    to another *device* path (e.g. three 1-D passes instead of the fused one)."""


_lock = threading.Lock()
_lib = None

_vp = ctypes.c_void_p
_i = ctypes.c_int
_d = ctypes.c_double
_sz = ctypes.c_size_t
_arr = ctypes.POINTER(MiArray)
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int32)

# name -> argtypes (restype is int unless listed in _RESTYPES); this table is
# also what tests/test_abi.py checks against include/mi355img.h
SIGNATURES = {
    "mi_version": [],
    "mi_last_error": [],
    "mi_device_count": [_ip],
    "mi_set_device": [_i],
    "mi_get_device": [_ip],
    "mi_device_name": [_i, ctypes.c_char_p, _sz],
    "mi_device_attr": [_i, _ip, _ip, ctypes.POINTER(_sz)],
    "mi_mem_info": [ctypes.POINTER(_sz), ctypes.POINTER(_sz)],
    "mi_malloc": [ctypes.POINTER(_vp), _sz],
    "mi_free": [_vp],
    "mi_pool_trim": [],
    "mi_pool_stats": [ctypes.POINTER(_sz), ctypes.POINTER(_sz)],
    "mi_memcpy_h2d": [_vp, _vp, _sz, _vp],
    "mi_memcpy_d2h": [_vp, _vp, _sz, _vp],
    "mi_memcpy_d2d": [_vp, _vp, _sz, _vp],
    "mi_memcpy_peer": [_vp, _i, _vp, _i, _sz, _vp],
    "mi_memset": [_vp, _i, _sz, _vp],
    "mi_stream_create": [ctypes.POINTER(_vp)],
    "mi_stream_destroy": [_vp],
    "mi_stream_sync": [_vp],
    "mi_default_stream": [ctypes.POINTER(_vp)],
    "mi_device_sync": [],
    "mi_event_create": [ctypes.POINTER(_vp)],
    "mi_event_destroy": [_vp],
    "mi_event_record": [_vp, _vp],
    "mi_stream_wait_event": [_vp, _vp],
    "mi_stream_wait_stream": [_vp, _vp],
    "mi_event_sync": [_vp],
    "mi_event_elapsed_ms": [_vp, _vp, ctypes.POINTER(ctypes.c_float)],
    "mi_copy": [_arr, _arr, _i, _vp],
    "mi_extend_rows": [_arr, _arr, _i, _i, _d, _vp],
    "mi_crop_rows": [_arr, _arr, _i, _vp],
    "mi_fill": [_arr, _d, _vp],
    "mi_any_diff": [_arr, _arr, _vp, _vp],
    "mi_elementwise": [_i, _arr, _arr, _arr, _vp],
    "mi_scalar_op": [_i, _arr, _arr, _d, _d, _d, _i, _vp],
    "mi_min_max": [_arr, _dp, _dp, _vp],
    "mi_sum": [_i, _arr, _arr, _dp, _vp],
    "mi_ssim_products": [_arr, _arr, _arr, _arr, _arr, _vp],
    "mi_ssim_combine_mean": [_arr, _arr, _arr, _arr, _arr, _arr, _i, _d, _d, _d, _dp, _vp],
    "mi_ssim_combine": [_arr] * 9 + [_d, _d, _d, _vp],
    "mi_correlate1d": [_arr, _arr, _i, _dp, _i, _i, _i, _d, _i, _vp],
    "mi_uniform_filter1d": [_arr, _arr, _i, _i, _i, _i, _d, _vp],
    "mi_separable3d_f32": [_arr, _arr, ctypes.POINTER(_dp), _ip, _ip, _ip, _d, _i, _vp],
    "mi_separable3d_f64": [_arr, _arr, ctypes.POINTER(_dp), _ip, _ip, _ip, _d, _vp],
    "mi_separable3d_f32_planes": [_arr, _arr, ctypes.POINTER(_dp), _ip, _ip, _ip, _d, _i64p, _i, _vp],
    "mi_separable3d_f32_supports": [_arr, _arr, ctypes.POINTER(_dp), _ip, _ip, _ip, _d, _i],
    "mi_correlate_nd": [_arr, _arr, _dp, _i64p, _ip, _i, _d, _i, _vp],
    "mi_correlate3_dense": [_arr, _arr, _dp, _i64p, _ip, _i, _d, _i, _vp],
    "mi_minmax1d": [_arr, _arr, _i, _i, _i, _i, _d, _i, _vp],
    "mi_minmax3d_u8": [_arr, _arr, _ip, _ip, _ip, _i, _i, _vp],
    "mi_minmax3d_f32": [_arr, _arr, _ip, _ip, _ip, _d, _i, _vp],
    "mi_minmax3d_f32_planes": [_arr, _arr, _ip, _ip, _ip, _d, _i, _i64p, _i, _vp],
    "mi_minmax3d_u8_planes": [_arr, _arr, _ip, _ip, _ip, _i, _i, _i64p, _i, _vp],
    "mi_minmax3d_16": [_arr, _arr, _ip, _ip, _ip, _i, _i, _vp],
    "mi_uniform2d_u8": [_arr, _arr, _ip, _i, _ip, _i, _vp],
    "mi_uniform2d_16": [_arr, _arr, _ip, _i, _ip, _i, _vp],
    "mi_uniform_z_u8": [_arr, _arr, _i, _i, _i, _i, _vp],
    "mi_uniform_z_16": [_arr, _arr, _i, _i, _i, _i, _vp],
    "mi_minmax_runs_u8": [_arr, _arr, _i, _ip, _ip, _i, _i, _vp],
    "mi_minmax_runs_16": [_arr, _arr, _i, _ip, _ip, _i, _i, _vp],
    "mi_minmax_runs_f32": [_arr, _arr, _i, _ip, _ip, _d, _i, _vp],
    "mi_minmax_runs3d_u8": [_arr, _arr, _ip, _ip, _i, _i, _vp],
    "mi_minmax3d_f64": [_arr, _arr, _ip, _ip, _ip, _d, _i, _vp],
    "mi_minmax_nd": [_arr, _arr, _u8p, _dp, _i64p, _ip, _i, _d, _i, _vp],
    "mi_rank_filter": [_arr, _arr, _u8p, _i64p, _ip, _i, _i, _d, _vp],
    "mi_median3x3": [_arr, _arr, _ip, _d, _vp],
    "mi_binary_erosion": [_arr, _arr, _u8p, _i64p, _ip, _arr, _i, _i, _vp, _vp],
    "mi_binary_erosion_fused": [_arr, _arr, _u8p, _i64p, _ip, _arr, _i, _i, _i, _vp, _vp],
    "mi_binary_open_close_fused": [_arr, _arr, _u8p, _i64p, _arr, _i, _i, _i, _vp],
    "mi_binary_propagation_step": [_arr, _arr, _u8p, _i64p, _ip, _arr, _i, _vp, _vp],
    "mi_map_coordinates": [_arr, _arr, _arr, _i, _i, _d, _vp],
    "mi_affine_transform": [_arr, _arr, _dp, _i, _i, _d, _vp],
    "mi_spline_pad": [_arr, _arr, _i, _i, _d, _vp],
    "mi_spline_filter1d": [_arr, _i, _i, _i, _vp],
    "mi_spline_prefilter": [_arr, _arr, _i, _i, _i, _i, _d, _vp],
    "mi_spline_map_coordinates": [_arr, _arr, _arr, _i, _i, _d, _i, _vp],
    "mi_spline_affine_transform": [_arr, _arr, _dp, _i, _i, _d, _i, _vp],
    "mi_comm_unique_id": [ctypes.c_char_p],
    "mi_comm_init_rank": [ctypes.POINTER(_vp), _i, _i, ctypes.c_char_p],
    "mi_comm_destroy": [_vp],
    "mi_halo_exchange": [_vp, _vp, _sz, ctypes.c_int64, _i, _i, _i, _i, _vp],
    "mi_comm_sendrecv": [_vp, _i, ctypes.POINTER(_vp), ctypes.POINTER(_sz), _ip, _ip, _vp],
    "mi_slab_separable3d_f32": [_vp, _arr, _arr, ctypes.POINTER(_dp), _ip, _ip, _ip, _d, _i, _i, _i, _i, _i,
                                _vp, _vp, _vp, _vp],
    "mi_slab_pipe_create": [ctypes.POINTER(_vp), _vp, _i, ctypes.POINTER(_arr), _arr, ctypes.POINTER(_dp), _ip, _ip, _ip, _d,
                            _i, _i, _i, _i, _vp],
    "mi_slab_pipe_destroy": [_vp],
    "mi_slab_pipe_step": [_vp, _i, _i],
    "mi_slab_pipe_run": [_vp, _i, _i],
    "mi_slab_pipe_info": [_vp, _ip, _ip, _ip],
}
_RESTYPES = {"mi_last_error": ctypes.c_char_p}


def library_path():
    """The in-tree library; MI355IMG_LIB names another build of it (A/B timing of kernel variants in separate
    processes -- a tuning aid, never a fallback: a path that does not load is an error)."""
    return os.environ.get("MI355IMG_LIB") or _build.LIB


def process_env_defaults():
    """Environment the HIP / HSA runtime must see when it initialises (its first call in the process), set with
    `setdefault` so that an explicit choice of the caller wins.  HSA_ENABLE_IPC_MODE_LEGACY=0: dmabuf IPC -- what RCCL
    between PROCESSES (one rank per GPU, distributed.HaloComm) needs on this driver; with the legacy mode the first
    send / recv fails with `hipIpcGetMemHandle: invalid argument`.  Harmless for single-process use.  Called by load()
    before the library (and with it the HIP runtime) is mapped; bench.py sets the same default at its top."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def load():
    """Return the loaded CDLL, building it first if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        process_env_defaults()
        path = library_path()
        if not os.path.exists(path):
            _build.build(verbose=False)
        try:
            lib = ctypes.CDLL(path)
        except OSError as exc:  # pragma: no cover - environment problem
            raise ImportError(
                "cupyimg_amd needs its HIP library {} (build it with "
                "`python -m cupyimg_amd._build`): {}".format(path, exc))
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, ctypes.c_int)
        _lib = lib
    return _lib


def last_error():
    msg = load().mi_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, exc=RuntimeError):
    """Map a C return code to a Python exception."""
    if rc == MI_OK:
        return
    msg = last_error()
    if rc == MI_ERR_UNSUPPORTED:
        raise Unsupported(msg)
    if rc == MI_ERR_NOMEM:
        raise MemoryError(msg)
    if rc == MI_ERR_INVALID_ARG:
        if "invalid origin" in msg or "output shape" in msg:
            raise ValueError(msg)
        raise exc(msg)
    if rc > 0:
        raise RuntimeError("HIP failure: " + msg)
    raise RuntimeError("libmi355img error {}: {}".format(rc, msg))
