"""NumPy front end of the CPU oracle (oracle/ndimage_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of ndimage_oracle.c.  The shipped
package ``cupyimg_amd`` never imports this module; tests, ``smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do.

The functions carry ``scipy.ndimage`` signatures so that parity tests read like
the reference's own tests (cupyimg/scipy/ndimage/tests/*).  Host-side argument
handling is restated independently of ``cupyimg_amd`` on purpose: two
implementations of the origin / flip / dtype rules have to agree with SciPy
and with each other.

Reference lines followed:
  filters.py:441-511 (correlate/convolve flip + origin rule), :549-665
  (uniform), :668-825 (gaussian), :1373-1557 (min/max); morphology.py:204-331,
  :396-461 (binary), :769-884 (grey); interpolation.py:271-394, :397-561.
Where the reference deviates from SciPy the
oracle follows SciPy 1.15.3, the parity target.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_MODES = {
    "reflect": 0, "grid-mirror": 0, "constant": 1, "nearest": 2, "mirror": 3,
    "wrap": 4, "grid-wrap": 5, "grid-constant": 6,
}
_DTYPES = ["bool", "int8", "uint8", "int16", "uint16", "int32", "uint32",
           "int64", "uint64", "float32", "float64"]

_i64p = ctypes.POINTER(ctypes.c_int64)
_intp = ctypes.POINTER(ctypes.c_int)
_dblp = ctypes.POINTER(ctypes.c_double)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile).  ORACLE_LIB=liboracle_asan.so in the environment selects the
    AddressSanitizer + UBSan build instead (tests/test_oracle_sanitizer.py; the process must have libasan preloaded)."""
    name = os.environ.get("ORACLE_LIB", "liboracle.so")
    if name not in ("liboracle.so", "liboracle_asan.so"):
        raise RuntimeError("ORACLE_LIB must be liboracle.so or liboracle_asan.so")
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "ndimage_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", name])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_boundary_index.restype = ctypes.c_int64
        _LIB.orc_boundary_index.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
    return _LIB


# ----------------------------------------------------------------- helpers
def _filter_mode(mode):
    """Filters: 'wrap' means grid-wrap, 'grid-constant' means constant
    (_filters_core.py:224-225)."""
    if mode not in _MODES:
        raise RuntimeError("boundary mode not supported (actual: {})".format(mode))
    code = _MODES[mode]
    return {4: 5, 6: 1}.get(code, code)


def _seq(arg, ndim, conv=lambda x: x):
    if isinstance(arg, str) or not hasattr(arg, "__iter__"):
        return [conv(arg)] * ndim
    lst = [conv(a) for a in arg]
    if len(lst) != ndim:
        raise RuntimeError("sequence argument must have length equal to input rank")
    return lst


def _shape_arr(shape):
    return (ctypes.c_int64 * max(len(shape), 1))(*shape)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, typ=_dblp):
    return a.ctypes.data_as(typ)


def _out_dtype(output, input):
    if output is None:
        return input.dtype
    if isinstance(output, np.ndarray):
        return output.dtype
    return np.dtype(output)


def cast(values_f64, dtype, round_half_even=False):
    """double -> dtype with C truncation semantics (orc_cast_from_f64)."""
    dtype = np.dtype(dtype)
    src = _f64(values_f64)
    dst = np.empty(src.shape, dtype)
    code = _DTYPES.index(dtype.name)
    rc = lib().orc_cast_from_f64(_ptr(src), dst.ctypes.data_as(ctypes.c_void_p),
                                 ctypes.c_int64(src.size), code, int(round_half_even))
    assert rc == 0
    return dst


def _finish(res_f64, output, input, round_half_even=False):
    dt = _out_dtype(output, input)
    res = cast(res_f64, dt, round_half_even)
    if isinstance(output, np.ndarray):
        if output.shape != res.shape:
            raise ValueError("output shape is not correct")
        output[...] = res
        return output
    return res


def boundary_index(i, n, mode):
    return int(lib().orc_boundary_index(int(i), int(n), _MODES[mode]))


# ------------------------------------------------------------- correlate
def _correlate1d_f64(x, w, axis, origin, mode, cval):
    x = _f64(x)
    w = _f64(w)
    out = np.empty_like(x)
    if x.ndim == 0 or x.size == 0:
        return x.copy()
    rc = lib().orc_correlate1d(_ptr(x), _ptr(out), _shape_arr(x.shape), x.ndim, axis,
                               _ptr(w), int(w.size), int(origin), _filter_mode(mode),
                               ctypes.c_double(cval))
    if rc == -2:
        raise ValueError("invalid origin")
    assert rc == 0, rc
    return out


def correlate1d(input, weights, axis=-1, output=None, mode="reflect", cval=0.0, origin=0):
    input = np.asarray(input)
    weights = np.asarray(weights, dtype=np.float64)
    if weights.ndim != 1 or weights.size < 1:
        raise RuntimeError("no filter weights given")
    axis = axis + input.ndim if axis < 0 else axis
    res = _correlate1d_f64(input, weights, axis, origin, mode, cval)
    return _finish(res, output, input)


def convolve1d(input, weights, axis=-1, output=None, mode="reflect", cval=0.0, origin=0):
    weights = np.asarray(weights)[::-1]
    origin = -origin
    if not len(weights) & 1:
        origin -= 1
    return correlate1d(input, weights, axis, output, mode, cval, origin)


def _correlate_nd_f64(x, w, origins, mode, cval):
    x = _f64(x)
    w = _f64(w)
    out = np.empty_like(x)
    if x.size == 0:
        return out
    org = (ctypes.c_int * x.ndim)(*[int(o) for o in origins])
    rc = lib().orc_correlate_nd(_ptr(x), _ptr(out), _shape_arr(x.shape), x.ndim, _ptr(w),
                                _shape_arr(w.shape), org, _filter_mode(mode),
                                ctypes.c_double(cval))
    if rc == -2:
        raise ValueError("invalid origin")
    assert rc == 0, rc
    return out


def correlate(input, weights, output=None, mode="reflect", cval=0.0, origin=0,
              _convolution=False):
    input = np.asarray(input)
    weights = np.asarray(weights, dtype=np.float64)
    if weights.ndim != input.ndim:
        raise RuntimeError("filter weights array has incorrect shape.")
    origins = _seq(origin, input.ndim, int)
    if _convolution:
        weights = weights[tuple([slice(None, None, -1)] * weights.ndim)]
        origins = [-o - (1 if s % 2 == 0 else 0) for o, s in zip(origins, weights.shape)]
    for o, s in zip(origins, weights.shape):
        if s // 2 + o < 0 or s // 2 + o >= s:
            raise ValueError("invalid origin")
    res = _correlate_nd_f64(input, weights, origins, mode, cval)
    return _finish(res, output, input)


def convolve(input, weights, output=None, mode="reflect", cval=0.0, origin=0):
    return correlate(input, weights, output, mode, cval, origin, _convolution=True)


# ------------------------------------------------------ uniform / gaussian
def uniform_filter1d(input, size, axis=-1, output=None, mode="reflect", cval=0.0, origin=0):
    input = np.asarray(input)
    if size < 1:
        raise RuntimeError("incorrect filter size")
    axis = axis + input.ndim if axis < 0 else axis
    x = _f64(input)
    out = np.empty_like(x)
    if x.size:
        rc = lib().orc_uniform1d(_ptr(x), _ptr(out), _shape_arr(x.shape), x.ndim, axis,
                                 int(size), int(origin), _filter_mode(mode),
                                 ctypes.c_double(cval))
        if rc == -2:
            raise ValueError("invalid origin")
        assert rc == 0, rc
    return _finish(out, output, input)


def uniform_filter(input, size=3, output=None, mode="reflect", cval=0.0, origin=0):
    input = np.asarray(input)
    dt = _out_dtype(output, input)
    sizes = _seq(size, input.ndim, int)
    origins = _seq(origin, input.ndim, int)
    modes = _seq(mode, input.ndim)
    cur = input
    did = False
    for ax in range(input.ndim):
        if sizes[ax] > 1:
            # every pass is stored in the output dtype (filters.py:651-662)
            cur = uniform_filter1d(cur, sizes[ax], ax, dt, modes[ax], cval, origins[ax])
            did = True
    if not did:
        cur = input.astype(dt)
    if isinstance(output, np.ndarray):
        output[...] = cur
        return output
    return cur


def gaussian_kernel1d(sigma, order, radius):
    """filters.py:795-825 (host-side, float64): Gaussian times the Hermite-like
    polynomial obtained from q_{k+1} = q_k' - x q_k / sigma^2."""
    if order < 0:
        raise ValueError("order must be non-negative")
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    phi = phi / phi.sum()
    if order == 0:
        return phi
    q = np.poly1d([1.0])
    mx = np.poly1d([-1.0 / (sigma * sigma), 0.0])
    for _ in range(order):
        q = q.deriv() + q * mx
    return q(x.astype(np.float64)) * phi


def gaussian_filter1d(input, sigma, axis=-1, order=0, output=None, mode="reflect",
                      cval=0.0, truncate=4.0):
    sd = float(sigma)
    lw = int(truncate * sd + 0.5)
    w = gaussian_kernel1d(sigma, order, lw)[::-1]
    return correlate1d(input, w, axis, output, mode, cval, 0)


def gaussian_filter(input, sigma, order=0, output=None, mode="reflect", cval=0.0,
                    truncate=4.0):
    input = np.asarray(input)
    dt = _out_dtype(output, input)
    sigmas = _seq(sigma, input.ndim, float)
    orders = _seq(order, input.ndim, int)
    modes = _seq(mode, input.ndim)
    cur = input
    did = False
    for ax in range(input.ndim):
        if sigmas[ax] > 1e-15:
            cur = gaussian_filter1d(cur, sigmas[ax], ax, orders[ax], dt, modes[ax], cval,
                                    truncate)
            did = True
    if not did:
        cur = input.astype(dt)
    if isinstance(output, np.ndarray):
        output[...] = cur
        return output
    return cur


# ---------------------------------------------------------------- min / max
def _minmax_nd_f64(x, fp, st, origins, mode, cval, is_max, typed=True):
    dcode = _DTYPES.index(np.asarray(x).dtype.name) if typed else -1
    x = _f64(x)
    out = np.empty_like(x)
    if x.size == 0:
        return out
    fp8 = np.ascontiguousarray(fp, dtype=np.uint8)
    stp = None if st is None else _f64(st)
    org = (ctypes.c_int * x.ndim)(*[int(o) for o in origins])
    rc = lib().orc_minmax_nd(_ptr(x), _ptr(out), _shape_arr(x.shape), x.ndim,
                             _ptr(fp8, _u8p), None if stp is None else _ptr(stp),
                             _shape_arr(fp8.shape), org, _filter_mode(mode),
                             ctypes.c_double(cval), int(is_max), dcode)
    if rc == -2:
        raise ValueError("invalid origin")
    assert rc == 0, rc
    return out


def _min_or_max_filter(input, size, footprint, structure, output, mode, cval, origin, is_max):
    input = np.asarray(input)
    if structure is None and footprint is None:
        if size is None:
            raise RuntimeError("no footprint or filter size provided")
        sizes = _seq(size, input.ndim, int)
        footprint = np.ones(sizes, bool)
    else:
        if footprint is not None:
            footprint = np.asarray(footprint, dtype=bool)
            if not footprint.any():
                raise ValueError("all-zero footprint is not supported")
        if structure is not None:
            structure = np.asarray(structure, dtype=np.float64)
            if footprint is None:
                footprint = np.ones(structure.shape, bool)
    if footprint.ndim != input.ndim:
        raise RuntimeError("footprint array has incorrect shape.")
    origins = _seq(origin, input.ndim, int)
    dt = _out_dtype(output, input)
    if structure is None and footprint.all():
        # separable: one 1-D pass per axis, each stored in the output dtype
        modes = _seq(mode, input.ndim)
        cur = input
        did = False
        for ax in range(input.ndim):
            if footprint.shape[ax] > 1:
                fshape = [1] * input.ndim
                fshape[ax] = footprint.shape[ax]
                org = [0] * input.ndim
                org[ax] = origins[ax]
                r = _minmax_nd_f64(cur, np.ones(fshape, bool), None, org, modes[ax], cval, is_max,
                                   typed=False)
                cur = cast(r, dt)
                did = True
        if not did:
            cur = input.astype(dt)
        if isinstance(output, np.ndarray):
            output[...] = cur
            return output
        return cur
    res = _minmax_nd_f64(input, footprint, structure, origins, mode, cval, is_max)
    return _finish(res, output, input)


def minimum_filter(input, size=None, footprint=None, output=None, mode="reflect", cval=0.0,
                   origin=0):
    return _min_or_max_filter(input, size, footprint, None, output, mode, cval, origin, False)


def maximum_filter(input, size=None, footprint=None, output=None, mode="reflect", cval=0.0,
                   origin=0):
    return _min_or_max_filter(input, size, footprint, None, output, mode, cval, origin, True)


def minimum_filter1d(input, size, axis=-1, output=None, mode="reflect", cval=0.0, origin=0):
    input = np.asarray(input)
    axis = axis + input.ndim if axis < 0 else axis
    fshape = [1] * input.ndim
    fshape[axis] = size
    org = [0] * input.ndim
    org[axis] = origin
    return _finish(_minmax_nd_f64(input, np.ones(fshape, bool), None, org, mode, cval, False,
                                  typed=False),
                   output, input)


def maximum_filter1d(input, size, axis=-1, output=None, mode="reflect", cval=0.0, origin=0):
    input = np.asarray(input)
    axis = axis + input.ndim if axis < 0 else axis
    fshape = [1] * input.ndim
    fshape[axis] = size
    org = [0] * input.ndim
    org[axis] = origin
    return _finish(_minmax_nd_f64(input, np.ones(fshape, bool), None, org, mode, cval, True,
                                  typed=False),
                   output, input)


def grey_erosion(input, size=None, footprint=None, structure=None, output=None,
                 mode="reflect", cval=0.0, origin=0):
    if size is None and footprint is None and structure is None:
        raise ValueError("size, footprint or structure must be specified")
    return _min_or_max_filter(input, size, footprint, structure, output, mode, cval, origin,
                              False)


def grey_dilation(input, size=None, footprint=None, structure=None, output=None,
                  mode="reflect", cval=0.0, origin=0):
    """morphology.py:818-884: max filter with everything mirrored."""
    if size is None and footprint is None and structure is None:
        raise ValueError("size, footprint or structure must be specified")
    input = np.asarray(input)
    flip = lambda a: a[tuple([slice(None, None, -1)] * a.ndim)]
    if structure is not None:
        structure = flip(np.asarray(structure))
    if footprint is not None:
        footprint = flip(np.asarray(footprint))
    origins = _seq(origin, input.ndim, int)
    for i in range(len(origins)):
        origins[i] = -origins[i]
        if footprint is not None:
            sz = footprint.shape[i]
        elif structure is not None:
            sz = structure.shape[i]
        elif np.isscalar(size):
            sz = size
        else:
            sz = size[i]
        if sz % 2 == 0:
            origins[i] -= 1
    return _min_or_max_filter(input, size, footprint, structure, output, mode, cval, origins,
                              True)


# ------------------------------------------------------------ binary morph
def generate_binary_structure(rank, connectivity):
    if connectivity < 1:
        connectivity = 1
    if rank < 1:
        return np.array(True, dtype=bool)
    out = np.fabs(np.indices([3] * rank) - 1)
    out = np.add.reduce(out, 0)
    return out <= connectivity


def _binary_erosion(input, structure, iterations, mask, output, border_value, origin, invert):
    input = np.asarray(input)
    if np.iscomplexobj(input):
        raise TypeError("Complex type not supported")
    if structure is None:
        structure = generate_binary_structure(input.ndim, 1)
    structure = np.ascontiguousarray(np.asarray(structure) != 0, dtype=np.uint8)
    if structure.ndim != input.ndim:
        raise RuntimeError("structure and input must have same dimensionality")
    if structure.size < 1:
        raise RuntimeError("structure must not be empty")
    if mask is not None:
        mask = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
        if mask.shape != input.shape:
            raise RuntimeError("mask and input must have equal sizes")
    origins = _seq(origin, input.ndim, int)
    cur = np.ascontiguousarray(input != 0, dtype=np.uint8)
    org = (ctypes.c_int * max(input.ndim, 1))(*origins)

    def one(src):
        dst = np.empty_like(src)
        if src.size:
            rc = lib().orc_binary_erosion(_ptr(src, _u8p), _ptr(dst, _u8p), _shape_arr(src.shape),
                                          src.ndim, _ptr(structure, _u8p),
                                          _shape_arr(structure.shape), org,
                                          None if mask is None else _ptr(mask, _u8p),
                                          int(bool(border_value)), int(invert))
            assert rc == 0, rc
        return dst

    it = 0
    while True:
        nxt = one(cur)
        it += 1
        changed = not np.array_equal(nxt, cur)
        cur = nxt
        if iterations >= 1 and it >= iterations:
            break
        if iterations < 1 and not changed:
            break
        if it > 100000:
            raise RuntimeError("binary morphology did not converge")
    res = cur.astype(bool)
    if isinstance(output, np.ndarray):
        output[...] = res
        return output
    return res if output is None else res.astype(output)


def binary_erosion(input, structure=None, iterations=1, mask=None, output=None,
                   border_value=0, origin=0, brute_force=False):
    return _binary_erosion(input, structure, iterations, mask, output, border_value, origin, 0)


def binary_dilation(input, structure=None, iterations=1, mask=None, output=None,
                    border_value=0, origin=0, brute_force=False):
    input = np.asarray(input)
    if structure is None:
        structure = generate_binary_structure(input.ndim, 1)
    structure = np.asarray(structure)
    origins = _seq(origin, input.ndim, int)
    structure = structure[tuple([slice(None, None, -1)] * structure.ndim)]
    for i in range(len(origins)):
        origins[i] = -origins[i]
        if not structure.shape[i] & 1:
            origins[i] -= 1
    return _binary_erosion(input, structure, iterations, mask, output, border_value, origins, 1)


# ------------------------------------------------------------ interpolation
def _interp_mode(mode):
    if mode not in _MODES:
        raise ValueError("boundary mode is not supported")
    return _MODES[mode]


def _spline_mode(mode):
    """spline boundary condition of the prefilter for an extension mode"""
    if mode in ("reflect", "grid-mirror", "nearest"):
        return 1
    if mode == "grid-wrap":
        return 2
    return 0


def spline_filter1d(input, order=3, axis=-1, output=np.float64, mode="mirror"):
    if order < 0 or order > 5:
        raise RuntimeError("spline order not supported")
    input = np.asarray(input)
    _interp_mode(mode)
    x = np.array(input, dtype=np.float64, order="C")
    if order > 1 and x.size and x.ndim:
        axis = axis % x.ndim
        rc = lib().orc_spline_filter1d(_ptr(x), _shape_arr(x.shape), x.ndim, int(axis), int(order), _spline_mode(mode))
        assert rc == 0, rc
    dt = np.dtype(output) if not isinstance(output, np.ndarray) else output.dtype
    res = x.astype(dt) if dt.kind == "f" else cast(x, dt)
    if isinstance(output, np.ndarray):
        output[...] = res
        return output
    return res


def spline_filter(input, order=3, output=np.float64, mode="mirror"):
    if order < 2 or order > 5:
        raise RuntimeError("spline order not supported")
    x = np.array(input, dtype=np.float64, order="C")
    for ax in range(x.ndim):
        x = spline_filter1d(x, order, ax, np.float64, mode)
    dt = np.dtype(output) if not isinstance(output, np.ndarray) else output.dtype
    res = x.astype(dt) if dt.kind == "f" else cast(x, dt)
    if isinstance(output, np.ndarray):
        output[...] = res
        return output
    return res


def _spline_coefficients(input, order, mode, cval, prefilter):
    """(float64 coefficients, npad): SciPy pads by 12 for nearest / grid-constant before prefiltering"""
    x = _f64(input)
    npad = 0
    if order > 1 and prefilter:
        if mode == "nearest":
            npad, x = 12, np.pad(x, 12, mode="edge")
        elif mode == "grid-constant":
            npad, x = 12, np.pad(x, 12, mode="constant", constant_values=cval)
        x = spline_filter(x, order, np.float64, mode)
    return np.ascontiguousarray(x), npad


def map_coordinates(input, coordinates, output=None, order=1, mode="constant", cval=0.0,
                    prefilter=True):
    if order < 0 or order > 5:
        raise RuntimeError("spline order not supported")
    input = np.asarray(input)
    coordinates = np.asarray(coordinates)
    x, npad = _spline_coefficients(input, order, mode, cval, prefilter)
    c = _f64(coordinates)
    oshape = c.shape[1:]
    nout = int(np.prod(oshape)) if oshape else 1
    out = np.empty(oshape, np.float64)
    if nout:
        rc = lib().orc_map_coordinates(_ptr(x), _shape_arr(x.shape), x.ndim, _ptr(c),
                                       ctypes.c_int64(nout), _ptr(out), int(order),
                                       _interp_mode(mode), ctypes.c_double(cval), int(npad))
        assert rc == 0, rc
    dt = _out_dtype(output, input)
    res = cast(out, dt, round_half_even=dt.kind in "iu")
    if isinstance(output, np.ndarray):
        output[...] = res
        return output
    return res


def _affine_args(input, matrix, offset):
    nd = input.ndim
    matrix = np.asarray(matrix, dtype=np.float64)
    if not hasattr(offset, "__iter__"):
        offset = [offset] * nd
    offset = np.asarray(offset, dtype=np.float64)
    if matrix.ndim == 2:
        if matrix.shape[0] == matrix.shape[1] - 1:
            offset = matrix[:, -1]
            matrix = matrix[:, :-1]
        elif matrix.shape[0] == nd + 1:
            offset = matrix[:-1, -1]
            matrix = matrix[:-1, :-1]
    elif matrix.ndim == 1:
        matrix = np.diag(matrix)
    else:
        raise RuntimeError("no proper affine matrix provided")
    m = np.zeros((nd, nd + 1))
    m[:, :nd] = matrix
    m[:, nd] = offset
    return m


def affine_transform(input, matrix, offset=0.0, output_shape=None, output=None, order=1,
                     mode="constant", cval=0.0, prefilter=True):
    if order < 0 or order > 5:
        raise RuntimeError("spline order not supported")
    input = np.asarray(input)
    nd = input.ndim
    m = _affine_args(input, matrix, offset)
    oshape = tuple(input.shape if output_shape is None else output_shape)
    x, npad = _spline_coefficients(input, order, mode, cval, prefilter)
    out = np.empty(oshape, np.float64)
    if out.size:
        rc = lib().orc_affine_transform(_ptr(x), _shape_arr(x.shape), nd, _ptr(_f64(m)),
                                        _ptr(out), _shape_arr(oshape), int(order),
                                        _interp_mode(mode), ctypes.c_double(cval), int(npad))
        assert rc == 0, rc
    dt = _out_dtype(output, input)
    res = cast(out, dt, round_half_even=dt.kind in "iu")
    if isinstance(output, np.ndarray):
        output[...] = res
        return output
    return res


def shift(input, shift, output=None, order=1, mode="constant", cval=0.0, prefilter=True):
    input = np.asarray(input)
    sh = _seq(shift, input.ndim, float)
    return affine_transform(input, np.ones(input.ndim), [-s for s in sh], None, output, order, mode, cval, prefilter)


def zoom(input, zoom, output=None, order=1, mode="constant", cval=0.0, prefilter=True, grid_mode=False):
    input = np.asarray(input)
    zf = _seq(zoom, input.ndim, float)
    oshape = tuple(int(round(n * z)) for n, z in zip(input.shape, zf))
    if grid_mode:
        scale = [n / o if o > 0 else 1.0 for n, o in zip(input.shape, oshape)]
        off = [0.5 * s - 0.5 for s in scale]
    else:
        scale = [(n - 1) / (o - 1) if o > 1 else 1.0 for n, o in zip(input.shape, oshape)]
        off = [0.0] * input.ndim
    return affine_transform(input, np.asarray(scale), off, oshape, output, order, mode, cval, prefilter)


# ---------------------------------------------------------- timed baselines
def uniform3d_f32(x, size, mode="reflect", cval=0.0):
    """Scalar single-thread CPU port of uniform_filter for float32 volumes
    (bench.py cpu_baseline)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    tmp = np.empty_like(x)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = lib().orc_uniform3d_f32(x.ctypes.data_as(fp), out.ctypes.data_as(fp),
                                 tmp.ctypes.data_as(fp), *[ctypes.c_int64(s) for s in x.shape],
                                 int(size), _filter_mode(mode), ctypes.c_double(cval))
    assert rc == 0
    return out


def minmax3d_u8(x, size, is_max, mode="reflect", cval=0):
    x = np.ascontiguousarray(x, dtype=np.uint8)
    out = np.empty_like(x)
    tmp = np.empty_like(x)
    rc = lib().orc_minmax3d_u8(_ptr(x, _u8p), _ptr(out, _u8p), _ptr(tmp, _u8p),
                               *[ctypes.c_int64(s) for s in x.shape], int(size),
                               _filter_mode(mode), int(cval), int(is_max))
    assert rc == 0
    return out
