"""scipy.ndimage interpolation on device arrays: spline_filter(1d),
map_coordinates, affine_transform, shift, zoom, rotate; spline orders 0-5.

Signatures follow cupyimg/scipy/ndimage/interpolation.py (spline_filter1d
:105-182, spline_filter :185-268, map_coordinates :271-394, affine_transform
:397-561, rotate :576-709, shift :712-802, zoom :805-990).  Orders 0 and 1 run
the gather kernels directly; orders 2-5 first build float64 B-spline
coefficients on the device (SciPy's rule: pad by 12 samples for `nearest` /
`grid-constant`, then prefilter every axis with the boundary condition that
matches the mode) and interpolate those -- any rank up to 8 (rank > 3: a plain
(order + 1)^rank tap loop in float64), results equal to SciPy 1.15's to rounding.
"""
import ctypes
import warnings

import numpy as np

from ... import core
from . import _support as S

__all__ = ["spline_filter1d", "spline_filter", "map_coordinates", "affine_transform", "shift", "zoom", "rotate"]

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
    return order


def _spline_mode_code(mode):
    """boundary condition of the prefilter for an extension mode (0 mirror, 1 reflect, 2 grid-wrap)"""
    if mode in ("reflect", "grid-mirror", "nearest"):
        return 1
    if mode == "grid-wrap":
        return 2
    return 0


_SPLINE_EXACT = 0x100      # csrc/interp.hip kSplExact


def _float32_route(src, out_dtype, order, allow_float32):
    """float32 image, float32 result, cubic spline: coefficients are stored as float32 and the
    gather kernel works in float32 (the reference's `allow_float32`, interpolation.py:330-335;
    about 1e-6 of the data range away from SciPy's double arithmetic)"""
    return (allow_float32 and order == 3 and src.dtype == np.float32 and np.dtype(out_dtype) == np.float32
            and src.ndim <= 3 and src.size < (1 << 28))


_IDENT_AXES = True              # test hook: False = every axis is filtered and interpolated (the route before round 5)
_SPLINE_SKIP_AXIS0 = 0x200      # csrc/interp.hip: spline_mode bit 9 + d leaves axis d unfiltered
_SPLINE_SAMPLES_AXIS0 = 0x100   # order bit 8 + d: axis d of the coefficient array holds samples


def _to_coefficients(input, order, mode, cval, prefilter, f32=False, exact=False, skip_axis=None):
    """(device array of B-spline coefficients, npad) for orders 2-5: float64, or float32
    on the float32 cubic route (the recursion itself always runs in double).
    SciPy pads by 12 samples for `nearest` / `grid-constant` before filtering;
    prefilter=False interpolates the samples as if they were coefficients.
    `exact`: only prefilter kernels whose arithmetic is SciPy's operation for operation (no blocked recursion) --
    set when the result is rounded to an integer dtype, where the last bit of a coefficient decides exact .5 ties."""
    src = core.ascontiguousarray(input)
    npad, pad_mode = 0, 0
    if prefilter and mode in ("nearest", "grid-constant"):
        npad, pad_mode = 12, (0 if mode == "nearest" else 1)
    if f32 and npad == 0 and not prefilter:
        return src, 0                      # the samples are the coefficients
    coef = core.empty(tuple(n + 2 * npad for n in src.shape), np.float32 if f32 else np.float64)
    a, b = src._desc(), coef._desc()
    lib = S.lib()
    if prefilter:
        S.check(lib.mi_spline_prefilter(ctypes.byref(a), ctypes.byref(b), int(order),
                                        _spline_mode_code(mode) | (_SPLINE_EXACT if exact else 0)
                                        | (0 if skip_axis is None else _SPLINE_SKIP_AXIS0 << skip_axis), npad, pad_mode, float(cval), None))
    else:
        S.check(lib.mi_spline_pad(ctypes.byref(a), ctypes.byref(b), npad, pad_mode, float(cval), None))
    return coef, npad


def _coef_dtype(input, ret, allow_float32):
    """coefficients are kept in float64 unless a float32 image is filtered into a float32 result
    with allow_float32 (interpolation.py:144-150); the recursion runs in double either way"""
    return np.float32 if (allow_float32 and input.dtype == np.float32 and ret.dtype == np.float32) else np.float64


def _spline_output(output, input):
    if isinstance(output, core.ndarray):
        if output.shape != input.shape:
            raise ValueError("output shape is not correct")
        return output
    return core.empty(input.shape, np.dtype(output))


def spline_filter1d(input, order=3, axis=-1, output=np.float64, mode="mirror", *, allow_float32=True):
    """B-spline prefilter along one axis (interpolation.py:105-182)."""
    if order < 0 or order > 5:
        raise RuntimeError("spline order not supported")
    input = S.as_device(input)
    if mode not in _INTERP_MODES:
        raise ValueError("boundary mode is not supported")
    ret = _spline_output(output, input)
    if order in (0, 1) or input.size == 0:
        ret[...] = input
        return ret
    axis = S.normalize_axis(axis, input.ndim)
    coef = core.ascontiguousarray(input).astype(_coef_dtype(input, ret, allow_float32))
    if core.shares_memory(coef, input):
        coef = coef.copy()
    d = coef._desc()
    if coef.shape[axis] > 1:
        S.check(S.lib().mi_spline_filter1d(ctypes.byref(d), axis, int(order), _spline_mode_code(mode), None))
    ret[...] = coef
    return ret


def spline_filter(input, order=3, output=np.float64, mode="mirror", *, allow_float32=True):
    """Multidimensional B-spline prefilter (interpolation.py:185-268)."""
    if order < 2 or order > 5:
        raise RuntimeError("spline order not supported")
    input = S.as_device(input)
    if mode not in _INTERP_MODES:
        raise ValueError("boundary mode is not supported")
    ret = _spline_output(output, input)
    if input.size == 0:
        return ret
    cdt = _coef_dtype(input, ret, allow_float32)
    src = core.ascontiguousarray(input)
    direct = ret._is_c_contiguous() and ret.dtype == cdt and not core.shares_memory(ret, src)
    coef = ret if direct else core.empty(src.shape, cdt)
    a, b = src._desc(), coef._desc()
    S.check(S.lib().mi_spline_prefilter(ctypes.byref(a), ctypes.byref(b), int(order), _spline_mode_code(mode), 0, 0, 0.0,
                                        None))
    if not direct:
        ret[...] = coef
    return ret


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
    if order > 1:
        coef, npad = _to_coefficients(src, order, mode, cval, prefilter,
                                      ret.ndim <= 3 and _float32_route(src, ret.dtype, order, allow_float32),
                                      exact=ret.dtype.kind in "iub")
        if coef is src and core.shares_memory(ret, src):
            coef = src.copy()
        ca_ = coef._desc()

        def launch_spline(dst):
            b = dst._desc()
            S.check(lib.mi_spline_map_coordinates(ctypes.byref(ca_), ctypes.byref(cd), ctypes.byref(b), order,
                                                  S.MODE_CODES[mode], float(cval), npad, None), ValueError)
        return _deliver(ret, launch_spline)

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
    return _affine(input, m, out, order, mode, cval, prefilter, allow_float32)


def _affine(input, m, out, order, mode, cval, prefilter, allow_float32=True):
    """out[o] = interp(input, m[:, :n] @ o + m[:, n]) -- the launch behind affine_transform / shift / zoom / rotate"""
    mk, mp = S.c_doubles(m)
    src = core.ascontiguousarray(input)
    lib = S.lib()
    if order > 1:
        f32 = _float32_route(src, out.dtype, order, allow_float32)
        exact = out.dtype.kind in "iub"
        # r5: an axis the matrix maps onto itself with an integral shift is evaluated at its samples, where the spline returns
        # them: its prefilter pass and its taps cancel (SciPy's own `rotate` filters the two axes of the rotation plane only).
        # The kernels that evaluate such an axis as ONE tap take it unfiltered (x: cubic3_rowblend_kernel -- `rotate` with the
        # default axes; the stream axis of cubic3_zfactor_kernel -- `rotate(axes=(1, 2))` / `(0, 2)`); anything else refuses
        # and the pass is made up for below.
        nd = src.ndim
        ident = None
        if _IDENT_AXES and f32 and prefilter and nd == 3 and order == 3 and out._is_c_contiguous() and min(src.shape) > 1:
            lin = m[:, :nd]
            if np.count_nonzero(lin - np.diag(np.diag(lin))):          # diagonal matrices take the separable resampling passes
                for d in (2, 0, 1):
                    row = np.zeros(nd)
                    row[d] = 1.0
                    if np.array_equal(lin[d], row) and float(m[d, nd]).is_integer() and abs(m[d, nd]) < 2 ** 20:
                        ident = d
                        break
        coef, npad = _to_coefficients(src, order, mode, cval, prefilter, f32, exact=exact, skip_axis=ident)
        if coef is src and core.shares_memory(out, src):
            coef = src.copy()
        ca_ = coef._desc()
        if ident is not None:
            b = out._desc()
            rc = lib.mi_spline_affine_transform(ctypes.byref(ca_), ctypes.byref(b), mp, order | (_SPLINE_SAMPLES_AXIS0 << ident),
                                                S.MODE_CODES[mode], float(cval), npad, None)
            if rc == 0:
                return out
            if rc != S._lib.MI_ERR_UNSUPPORTED:
                S.check(rc, ValueError)
            # no single-tap kernel for this matrix: filter the axis after all (the passes commute)
            S.check(lib.mi_spline_filter1d(ctypes.byref(ca_), ident, int(order), _spline_mode_code(mode), None))

        def launch_spline(dst):
            b = dst._desc()
            S.check(lib.mi_spline_affine_transform(ctypes.byref(ca_), ctypes.byref(b), mp, order, S.MODE_CODES[mode],
                                                   float(cval), npad, None), ValueError)
        return _deliver(out, launch_spline)

    def launch(dst):
        def fn(s):
            a, b = s._desc(), dst._desc()
            S.check(lib.mi_affine_transform(ctypes.byref(a), ctypes.byref(b), mp, order, S.MODE_CODES[mode],
                                            float(cval), None), ValueError)
        _call_with_rank_fallback(src, fn)

    if core.shares_memory(out, src):
        src = src.copy()
    return _deliver(out, launch)


def shift(input, shift, output=None, order=3, mode="constant", cval=0.0, prefilter=True, *, allow_float32=True):
    """Shift an array (interpolation.py:712-802): output[o] = input[o - shift]."""
    order = _check_parameter("shift", order, mode)
    input = S.as_device(input)
    nd = input.ndim
    sh = [float(v) for v in S.normalize_sequence(shift, nd)]
    out = _get_output(output, input, input.shape)
    if out.size == 0:
        return out
    m = np.zeros((nd, nd + 1))
    m[:, :nd] = np.eye(nd)
    m[:, nd] = [-v for v in sh]
    return _affine(input, m, out, order, mode, cval, prefilter, allow_float32)


def zoom(input, zoom, output=None, order=3, mode="constant", cval=0.0, prefilter=True, *, grid_mode=False,
         allow_float32=True):
    """Zoom an array (interpolation.py:805-990).  Output extents are
    round(n * zoom); without grid_mode the corner samples map onto each other
    (scale (n - 1) / (m - 1)), with it pixel edges do (scale n / m)."""
    order = _check_parameter("zoom", order, mode)
    input = S.as_device(input)
    nd = input.ndim
    zf = [float(v) for v in S.normalize_sequence(zoom, nd)]
    oshape = tuple(int(round(n * z)) for n, z in zip(input.shape, zf))
    if grid_mode:
        if mode in ("constant", "wrap"):
            warnings.warn("It is recommended to use mode = grid-{0} instead of {0} when grid_mode is True.".format(mode),
                          stacklevel=2)
        scale = [n / o if o > 0 else 1.0 for n, o in zip(input.shape, oshape)]
        off = [0.5 * s - 0.5 for s in scale]
    else:
        scale = [(n - 1) / (o - 1) if o > 1 else 1.0 for n, o in zip(input.shape, oshape)]
        off = [0.0] * nd
    out = _get_output(output, input, oshape)
    if out.size == 0:
        return out
    m = np.zeros((nd, nd + 1))
    m[:, :nd] = np.diag(scale)
    m[:, nd] = off
    return _affine(input, m, out, order, mode, cval, prefilter, allow_float32)


def _cos_sin_deg(angle):
    """cos / sin of an angle in degrees, exact at multiples of 90 (like scipy.special.cosdg / sindg)"""
    a = float(angle) % 360.0
    if a % 90.0 == 0.0:
        return [(1.0, 0.0), (0.0, 1.0), (-1.0, 0.0), (0.0, -1.0)][int(a // 90) % 4]
    r = np.deg2rad(a)
    return float(np.cos(r)), float(np.sin(r))


def rotate(input, angle, axes=(1, 0), reshape=True, output=None, order=3, mode="constant", cval=0.0,
           prefilter=True, *, allow_float32=True):
    """Rotate an array in the plane of two axes (interpolation.py:576-709)."""
    order = _check_parameter("rotate", order, mode)
    input = S.as_device(input)
    nd = input.ndim
    if nd < 2:
        raise ValueError("input array should be at least 2D")
    axes = list(axes)
    if len(axes) != 2:
        raise ValueError("axes should contain exactly two values")
    if not all(float(ax).is_integer() for ax in axes):
        raise ValueError("axes should contain only integer values")
    axes = [int(ax) + nd if ax < 0 else int(ax) for ax in axes]
    if axes[0] >= nd or axes[1] >= nd or axes[0] < 0 or axes[1] < 0:
        raise ValueError("invalid rotation plane specified")
    axes.sort()
    c, s = _cos_sin_deg(angle)
    rot = np.array([[c, s], [-s, c]])
    img_shape = np.asarray(input.shape)
    in_plane = img_shape[axes]
    if reshape:
        iy, ix = in_plane
        bounds = rot @ np.array([[0, 0, iy, iy], [0, ix, 0, ix]], dtype=np.float64)
        out_plane = (np.ptp(bounds, axis=1) + 0.5).astype(int)
    else:
        out_plane = img_shape[axes]
    out_center = rot @ ((out_plane - 1) / 2.0)
    in_center = (in_plane - 1) / 2.0
    offset = in_center - out_center
    oshape = img_shape.copy()
    oshape[axes] = out_plane
    out = _get_output(output, input, tuple(int(v) for v in oshape))
    if out.size == 0:
        return out
    # identity on the other axes: one launch instead of SciPy's loop over planes (the samples on those axes sit
    # at integral coordinates, where the spline reproduces them)
    m = np.zeros((nd, nd + 1))
    m[:, :nd] = np.eye(nd)
    m[axes[0], axes[0]], m[axes[0], axes[1]] = rot[0]
    m[axes[1], axes[0]], m[axes[1], axes[1]] = rot[1]
    m[axes[0], nd], m[axes[1], nd] = offset
    return _affine(input, m, out, order, mode, cval, prefilter, allow_float32)
