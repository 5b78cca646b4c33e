"""Host-side argument handling shared by the ndimage API functions.

Behaviour (accepted values, defaults, exception types) follows
cupyimg/scipy/ndimage/_util.py and _filters_core.py:10-109 so the parity
tests can assert the same errors the reference's tests do; the code is
organised around the C-ABI rather than around CuPy kernels.
"""
import contextlib
import ctypes
import functools
import inspect
import threading

import numpy as np

from ... import _lib, core
from ..._lib import MODE_CODES

_FILTER_MODES = ("reflect", "constant", "nearest", "mirror", "wrap", "grid-mirror", "grid-wrap",
                 "grid-constant")


def check_mode(mode):
    """_util.py:105-119"""
    if mode not in _FILTER_MODES:
        raise RuntimeError("boundary mode not supported (actual: {})".format(mode))
    return mode


def mode_code(mode):
    return MODE_CODES[check_mode(mode)]


def check_origin(origin, width):
    """_util.py:98-102"""
    origin = int(origin)
    if (width // 2 + origin < 0) or (width // 2 + origin >= width):
        raise ValueError("invalid origin")
    return origin


def fix_sequence_arg(arg, ndim, name, conv=lambda x: x):
    """_util.py:84-95"""
    if isinstance(arg, str):
        return [conv(arg)] * ndim
    try:
        arg = iter(arg)
    except TypeError:
        return [conv(arg)] * ndim
    lst = [conv(x) for x in arg]
    if len(lst) != ndim:
        raise RuntimeError("{} must have length equal to input rank".format(name))
    return lst


def normalize_sequence(arr, rank):
    """_util.py:137-151"""
    if isinstance(arr, core.ndarray):
        arr = arr.get()
    if hasattr(arr, "__iter__") and not isinstance(arr, str):
        normalized = list(arr)
        if len(normalized) != rank:
            raise RuntimeError("sequence argument must have length equal to arr rank")
    else:
        normalized = [arr] * rank
    return normalized


def normalize_axis(axis, ndim):
    axis = int(axis)
    if axis < -ndim or axis >= ndim:
        raise ValueError("axis {} is out of bounds for array of dimension {}".format(axis, ndim))
    return axis % ndim if ndim else 0


def as_device(a, name="input"):
    """Device array from a device array or anything array-like on the host."""
    if isinstance(a, core.ndarray):
        return a
    arr = np.asarray(a)
    if arr.dtype.kind == "c":
        raise TypeError("Complex type not supported")
    if arr.dtype == np.float16:      # reached only by functions without `output` (float16_aware handles the rest)
        arr = arr.astype(np.float32)
    return core.asarray(arr)


_F16 = np.dtype(np.float16)


def _is_f16(a):
    if isinstance(a, (core.ndarray, np.ndarray)):
        return a.dtype == _F16
    if isinstance(a, (type, np.dtype, str)):
        try:
            return np.dtype(a) == _F16
        except TypeError:
            return False
    return False


def float16_aware(fn):
    """float16 images keep their dtype, as in the reference (_filters_core.py:169-171: float16 in, float16 out, the
    arithmetic in float32 / float64): the kernels of this library have no float16 arithmetic, so a float16 input is
    converted to float32 on the device (exact), the call runs in float32, and the result is rounded to float16 when
    the caller asked for float16 (explicitly, or by passing a float16 image without `output`).  Calls that involve no
    float16 pass straight through (the test costs a few attribute look-ups)."""
    sig = inspect.signature(fn)
    if "output" not in sig.parameters:
        return fn
    first = next(iter(sig.parameters))

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if not (any(_is_f16(a) for a in args) or any(_is_f16(v) for v in kwargs.values())):
            return fn(*args, **kwargs)
        bound = sig.bind(*args, **kwargs)
        ba = bound.arguments
        image = ba[first]
        in16 = _is_f16(image)
        if in16:
            ba[first] = image.astype(np.float32)
        out = ba.get("output", None)
        out_arr = out if isinstance(out, core.ndarray) and _is_f16(out) else None
        out16 = out_arr is not None or (out is not None and not isinstance(out, core.ndarray) and _is_f16(out))
        if out16:
            ba["output"] = np.float32
        res = fn(*bound.args, **bound.kwargs)
        if not isinstance(res, core.ndarray):
            return res
        if out_arr is not None:
            out_arr[...] = res
            return out_arr
        # no `output` given: functions whose result takes the input dtype (now float32) give float16 back; the
        # others (bool masks of binary morphology, float64 spline coefficients) keep their own default
        if out16 or (out is None and in16 and res.dtype == np.float32):
            return res.astype(np.float16)
        return res
    return wrapper


def as_host(a, dtype=None):
    """Small parameter arrays (weights, footprints, matrices) live on the host:
    they are tiny and the reference's device syncs on them
    (_filters_core.py:35,39; morphology.py:133,274) are avoided that way."""
    if isinstance(a, core.ndarray):
        a = core.host_copy(a)
    return np.asarray(a, dtype=dtype)


def is_integer_output(output, input):
    """_util.py:4-9"""
    if output is None:
        return input.dtype.kind in "iu"
    if isinstance(output, core.ndarray):
        return output.dtype.kind in "iu"
    return np.dtype(output).kind in "iu"


def check_cval(mode, cval, integer_output):
    """_util.py:12-17"""
    if mode == "constant" and integer_output and not np.isfinite(cval):
        raise NotImplementedError("Non-finite cval is not supported for outputs with integer dtype.")


def get_output(output, input, shape=None):
    """Resolve the `output` argument (_util.py:43-81): a device array is
    shape-checked and used as is; a dtype or None allocates.  Unlike the
    reference nothing is zero-filled -- every kernel writes every element."""
    if shape is None:
        shape = input.shape
    if isinstance(output, core.ndarray):
        if output.shape != tuple(shape):
            raise ValueError("output shape is not correct")
        output._touch()            # about to be overwritten: a remembered host copy (core.with_host_hint) is stale
        return output
    if isinstance(output, np.ndarray):
        raise TypeError("output must be a device array (cupyimg_amd.ndarray) or a dtype")
    dtype = input.dtype if output is None else np.dtype(output)
    if dtype.kind == "c":
        raise TypeError("Complex type not supported")
    return core.empty(shape, dtype)


def acc_flag(dtype_mode):
    if dtype_mode == "ndimage":
        return 0
    if dtype_mode == "float":
        return 1
    raise ValueError("dtype_mode={} is not supported".format(dtype_mode))


# Plane-restricted launches (multi-GPU slabs, distributed.SlabFilter): inside
# `output_planes([(b0, e0), (b1, e1)])` a filter call computes only those
# output planes along axis 0.  Only the fused 3-D kernel honours it; every
# other path raises Unsupported rather than silently filtering everything.
_scope = threading.local()


@contextlib.contextmanager
def output_planes(ranges):
    prev = getattr(_scope, "planes", None)
    _scope.planes = [(int(b), int(e)) for b, e in ranges]
    try:
        yield
    finally:
        _scope.planes = prev


def current_planes():
    return getattr(_scope, "planes", None)


def _no_plane_scope():
    if current_planes() is not None:
        raise Unsupported("this filter cannot be restricted to a range of output planes")


def run_kernel(input, output, launch):
    """Run ``launch(src, dst)`` with contiguous, non-overlapping src/dst and
    deliver the result into `output` (which may be a strided view or alias the
    input; the reference handles the latter with a temp + copy,
    _filters_core.py:148-155)."""
    _no_plane_scope()
    src = core.ascontiguousarray(input)
    if output._is_c_contiguous() and not core.shares_memory(output, src):
        launch(src, output)
        return output
    tmp = core.empty(output.shape, output.dtype)
    launch(src, tmp)
    output[...] = tmp
    return output


def run_passes(input, output, passes):
    """Chain of 1-D passes ``f(src, dst)`` ping-ponging between `output` and
    one scratch volume so that the last pass lands in `output` and no
    copy-back is needed (the reference pays a zero-fill plus a full copy per
    in-place pass, _filters_core.py:148-155; filters.py:651-662)."""
    _no_plane_scope()
    n = len(passes)
    if n == 0:
        output[...] = input
        return output
    src = core.ascontiguousarray(input)
    direct = output._is_c_contiguous()
    final = output if direct else core.empty(output.shape, output.dtype)
    if core.shares_memory(final, src):
        src = src.copy()
    scratch = core.empty(output.shape, output.dtype) if n > 1 else None
    for i, f in enumerate(passes):
        dst = final if (n - 1 - i) % 2 == 0 else scratch
        f(src, dst)
        src = dst
    if not direct:
        output[...] = final
    return output


_EW_OPS = {"add": 0, "subtract": 1, "multiply": 2, "sqrt": 3}


def elementwise(op, a, b, out):
    """out[...] = a (op) b in out's dtype (operands of another dtype are cast
    first, like a NumPy ufunc with `out=`); integers wrap.  mi_elementwise."""
    a = core.ascontiguousarray(a if a.dtype == out.dtype else a.astype(out.dtype))
    if b is not None:
        b = core.ascontiguousarray(b if b.dtype == out.dtype else b.astype(out.dtype))
    dst = out if out._is_c_contiguous() else core.empty(out.shape, out.dtype)
    da, dd = a._desc(), dst._desc()
    if b is None:
        check(lib().mi_elementwise(_EW_OPS[op], ctypes.byref(da), None, ctypes.byref(dd), None))
    else:
        db = b._desc()
        check(lib().mi_elementwise(_EW_OPS[op], ctypes.byref(da), ctypes.byref(db), ctypes.byref(dd), None))
    if dst is not out:
        out[...] = dst
    return out


def scale_shift(a, scale, shift, dtype=None):
    """a * scale + shift for a float device array (mi_scalar_op); with `dtype` (float32 / float64) an 8- / 16-bit
    integer array is converted and scaled in the same pass"""
    a = core.ascontiguousarray(a)
    out = core.empty(a.shape, a.dtype if dtype is None else dtype)
    da, dd = a._desc(), out._desc()
    check(lib().mi_scalar_op(0, ctypes.byref(da), ctypes.byref(dd), float(scale), float(shift), 0.0, 0, None))
    return out


def clip(a, lo, hi, keep=None):
    """clip a float device array to [lo, hi]; entries equal to `keep` stay"""
    a = core.ascontiguousarray(a)
    out = core.empty(a.shape, a.dtype)
    da, dd = a._desc(), out._desc()
    check(lib().mi_scalar_op(1, ctypes.byref(da), ctypes.byref(dd), float(lo), float(hi),
                             0.0 if keep is None else float(keep), int(keep is not None), None))
    return out


def min_max(a):
    """(min, max) of a device array as Python floats (one small read-back)"""
    a = core.ascontiguousarray(a)
    lo, hi = ctypes.c_double(), ctypes.c_double()
    da = a._desc()
    check(lib().mi_min_max(ctypes.byref(da), ctypes.byref(lo), ctypes.byref(hi), None))
    return lo.value, hi.value


def c_doubles(values):
    arr = np.ascontiguousarray(values, dtype=np.float64)
    return arr, arr.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def c_ints(values):
    return (ctypes.c_int * max(len(values), 1))(*[int(v) for v in values])


def c_int64s(values):
    return (ctypes.c_int64 * max(len(values), 1))(*[int(v) for v in values])


def lib():
    return _lib.load()


check = _lib.check
MODE_CODES = MODE_CODES
Unsupported = _lib.Unsupported
