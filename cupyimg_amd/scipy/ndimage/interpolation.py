"""scipy.ndimage interpolation on device arrays: map_coordinates and
affine_transform, spline orders 0 and 1.

Signatures follow cupyimg/scipy/ndimage/interpolation.py (map_coordinates
:271-394, affine_transform :397-561).  Orders 2-5 need the spline prefilter
(_spline_prefilter_core.py), which is the first "next" item of the scope table
and not built yet: they raise NotImplementedError instead of silently using a
different algorithm.
"""
import ctypes
import warnings

import numpy as np

from ... import core
from . import _support as S

__all__ = ["map_coordinates", "affine_transform"]

_INTERP_MODES = ("constant", "grid-constant", "nearest", "mirror", "reflect", "grid-mirror", "wrap",
                 "grid-wrap")


def _check_parameter(func_name, order, mode):
    """interpolation.py:45-60"""
    if order is None:
        order = 1
    if order < 0 or 5 < order:
        raise ValueError("spline order is not supported")
    if mode in ("opencv", "_opencv_edge"):
        raise NotImplementedError("the 'opencv' pseudo-modes are not part of the scipy.ndimage API")
    if mode not in _INTERP_MODES:
        raise ValueError("boundary mode is not supported")
    if order > 1:
        raise NotImplementedError(
            "{}: spline order {} needs the B-spline prefilter, which is not built yet "
            "(orders 0 and 1 are)".format(func_name, order))
    return order


def _get_output(output, input, shape):
    """interpolation.py:31-42"""
    if isinstance(output, core.ndarray):
        if output.shape != tuple(shape):
            raise ValueError("output shape is not correct")
        return output
    dtype = input.dtype if output is None else np.dtype(output)
    return core.empty(shape, dtype)


def _deliver(output, launch):
    if output._is_c_contiguous():
        launch(output)
        return output
    tmp = core.empty(output.shape, output.dtype)
    launch(tmp)
    output[...] = tmp
    return output


def _call_with_rank_fallback(src, fn):
    """rank > 3 kernels exist for float input only; other dtypes are converted
    to float64 first, which is exact (SciPy interpolates in double anyway)."""
    try:
        fn(src)
    except S.Unsupported:
        fn(src.astype(np.float64))


def map_coordinates(input, coordinates, output=None, order=3, mode="constant", cval=0.0,
                    prefilter=True, *, allow_float32=True):
    """Map the input array to new coordinates by interpolation
    (interpolation.py:271-394).  ``coordinates`` has shape (ndim, *out_shape)."""
    order = _check_parameter("map_coordinates", order, mode)
    input = S.as_device(input)
    if isinstance(coordinates, core.ndarray):
        coords = coordinates
        ckind = coords.dtype.kind
    else:
        coords = np.asarray(coordinates)
        ckind = coords.dtype.kind
    if ckind in "iub":
        coords = coords.astype(np.float64)
    elif ckind != "f":
        raise ValueError("coordinates should have floating point dtype")
    if not isinstance(coords, core.ndarray):
        if coords.dtype == np.float16:
            coords = coords.astype(np.float32)
        coords = core.asarray(coords)
    if coords.ndim < 1 or coords.shape[0] != input.ndim:
        raise RuntimeError("invalid shape for coordinate array")
    ret = _get_output(output, input, coords.shape[1:])
    if ret.size == 0:
        return ret
    src = core.ascontiguousarray(input)
    coords = core.ascontiguousarray(coords)
    cd = coords._desc()
    lib = S.lib()

    def launch(dst):
        def fn(s):
            a, b = s._desc(), dst._desc()
            S.check(lib.mi_map_coordinates(ctypes.byref(a), ctypes.byref(cd), ctypes.byref(b), order,
                                           S.MODE_CODES[mode], float(cval), None), ValueError)
        _call_with_rank_fallback(src, fn)

    return _deliver(ret, launch)


def affine_transform(input, matrix, offset=0.0, output_shape=None, output=None, order=3,
                     mode="constant", cval=0.0, prefilter=True, *, allow_float32=True):
    """Apply an affine transformation (interpolation.py:397-561): output voxel
    ``o`` samples the input at ``matrix @ o + offset``."""
    order = _check_parameter("affine_transform", order, mode)
    input = S.as_device(input)
    ndim = input.ndim
    if not hasattr(offset, "__iter__") and not isinstance(offset, core.ndarray):
        offset = [offset] * ndim
    offset = S.as_host(offset, np.float64)
    matrix = S.as_host(matrix, np.float64)
    if matrix.ndim not in (1, 2):
        raise RuntimeError("no proper affine matrix provided")
    if matrix.ndim == 2:
        if matrix.shape[0] == matrix.shape[1] - 1:
            offset = matrix[:, -1]
            matrix = matrix[:, :-1]
        elif matrix.shape[0] == ndim + 1:
            offset = matrix[:-1, -1]
            matrix = matrix[:-1, :-1]
        if matrix.shape != (ndim, ndim):
            raise RuntimeError("improper affine shape")
    else:
        if matrix.shape[0] != ndim:
            raise RuntimeError("improper affine shape")
        warnings.warn("The behavior of affine_transform with a 1-D array supplied for the matrix "
                      "parameter has changed in SciPy 0.18.0.", stacklevel=2)
        # diagonal form == zoom + shift (interpolation.py:532-545)
        matrix = np.diag(matrix)
    if offset.shape != (ndim,):
        raise RuntimeError("offset must have length equal to input rank")
    if output_shape is None:
        output_shape = output.shape if isinstance(output, core.ndarray) else input.shape
    if len(output_shape) != ndim:
        raise RuntimeError("output_shape must have length equal to input rank")
    out = _get_output(output, input, tuple(output_shape))
    if out.size == 0:
        return out
    m = np.zeros((ndim, ndim + 1), dtype=np.float64)
    m[:, :ndim] = matrix
    m[:, ndim] = offset
    mk, mp = S.c_doubles(m)
    src = core.ascontiguousarray(input)
    lib = S.lib()

    def launch(dst):
        def fn(s):
            a, b = s._desc(), dst._desc()
            S.check(lib.mi_affine_transform(ctypes.byref(a), ctypes.byref(b), mp, order, S.MODE_CODES[mode],
                                            float(cval), None), ValueError)
        _call_with_rank_fallback(src, fn)

    if core.shares_memory(out, src):
        src = src.copy()
    return _deliver(out, launch)
