"""scipy.signal direct convolution callers of the n-D correlate kernel
(cupyimg/scipy/signal/signaltools.py:71-180 `_convolveND` / `_correlateND`, and the public
convolve / correlate / convolve2d / correlate2d built on them; `dtype_mode="numpy"` at
scipy/ndimage/filters.py:470-487).

Only the direct method exists here: it is the one that runs on the filtering path
(`method="auto"` resolves to it; `method="fft"` is outside this package).  Results have
NumPy's promoted dtype, not ndimage's float policy."""
import numpy as np

from ... import core, _pad
from .. import ndimage as ndi

__all__ = ["convolve", "correlate", "convolve2d", "correlate2d"]

_NDI_BOUNDARY = {"fill": "constant", "pad": "constant", "wrap": "grid-wrap", "circular": "grid-wrap", "symm": "reflect",
                 "symmetric": "reflect"}
_PAD_BOUNDARY = {"fill": "constant", "pad": "constant", "wrap": "wrap", "circular": "wrap", "symm": "symmetric",
                 "symmetric": "symmetric"}


def _dev(a):
    return a if isinstance(a, core.ndarray) else core.asarray(np.asarray(a))


def _host(a):
    return a.get() if isinstance(a, core.ndarray) else np.asarray(a)


def _inputs_swap_needed(mode, shape1, shape2):
    """'valid' needs the larger operand first (signaltools.py:183-220)"""
    if mode != "valid":
        return False
    ok1 = all(a >= b for a, b in zip(shape1, shape2))
    ok2 = all(b >= a for a, b in zip(shape1, shape2))
    if not (ok1 or ok2):
        raise ValueError("For 'valid' mode, one must be at least as large as the other in every dimension")
    return not ok1


def _reverse(a):
    return a[(slice(None, None, -1),) * a.ndim]


def _convolve_nd(in1, kernel, mode, boundary="fill", fillvalue=0, corr2d=False):
    """linear convolution of the device array `in1` with the host array `kernel`:
    pad for 'full', run the correlate kernel with the reversed weights (origin 0 lands on
    SciPy's centring for odd and even lengths), slice for 'valid' (signaltools.py:71-147)"""
    if not np.isscalar(fillvalue):
        raise ValueError("non-scalar fillvalue not supported")
    if boundary not in _NDI_BOUNDARY:
        raise ValueError("Acceptable boundary flags are 'fill', 'circular' (or 'wrap'), and 'symmetric' (or 'symm').")
    if mode not in ("full", "same", "valid"):
        raise ValueError("Acceptable mode flags are 'valid', 'same', or 'full'.")
    sizes = kernel.shape
    if mode == "full":
        if _PAD_BOUNDARY[boundary] == "constant":
            in1 = _pad.pad(in1, [((s - 1) // 2, s - 1 - (s - 1) // 2) for s in sizes], "constant", fillvalue)
            crop = None
        else:
            in1 = _pad.pad(in1, [(s - 1, s - 1) for s in sizes], _PAD_BOUNDARY[boundary])
            crop = tuple(slice((s - 1) - (s - 1) // 2, n - (s - 1) + (s - 1 - (s - 1) // 2))
                         for s, n in zip(sizes, in1.shape))
    # correlate2d centres 'same' one sample later than convolve2d for even lengths (SciPy's 2-D routine
    # without the kernel flip); everything else is centred like `_centered(full, in1.shape)`
    origin = [-1 if (corr2d and mode == "same" and s % 2 == 0) else 0 for s in sizes]
    out = ndi.correlate(in1, _reverse(kernel), mode=_NDI_BOUNDARY[boundary], cval=fillvalue, origin=origin,
                        dtype_mode="numpy")
    if mode == "valid":
        crop = tuple(slice(s - 1 - (s - 1) // 2, n - (s - 1) // 2) for s, n in zip(sizes, out.shape))
    elif mode == "same":
        crop = None
    return core.ascontiguousarray(out[crop]) if crop is not None else out


def _check(in1, in2):
    if in1.ndim == in2.ndim == 0:
        raise ValueError("0-d inputs are not supported")
    if in1.ndim != in2.ndim:
        raise ValueError("in1 and in2 should have the same dimensionality")
    if in1.dtype.kind == "c" or in2.dtype.kind == "c":
        raise NotImplementedError("complex arrays are outside the filtering path")


def _method(method):
    if method not in ("auto", "direct", "fft"):
        raise ValueError("Acceptable method flags are 'auto', 'direct', or 'fft'.")
    if method == "fft":
        raise NotImplementedError("only the direct method runs on the filtering path")


def convolve(in1, in2, mode="full", method="auto"):
    """Convolve two N-dimensional arrays, direct method (signaltools.py `convolve`)."""
    _method(method)
    a, b = _dev(in1), in2
    _check(a, b if hasattr(b, "ndim") else np.asarray(b))
    if _inputs_swap_needed(mode, a.shape, np.shape(_host(b)) if not isinstance(b, core.ndarray) else b.shape):
        a, b = _dev(in2), in1
    return _convolve_nd(a, _host(b), mode)


def correlate(in1, in2, mode="full", method="auto"):
    """Cross-correlate two N-dimensional arrays, direct method (signaltools.py `correlate`)."""
    _method(method)
    a, b = _dev(in1), in2
    _check(a, b if hasattr(b, "ndim") else np.asarray(b))
    bshape = b.shape if isinstance(b, core.ndarray) else np.shape(_host(b))
    if _inputs_swap_needed(mode, a.shape, bshape):
        # correlate(x, y)[k] = correlate(y, x)[-k]
        out = _convolve_nd(_dev(in2), _reverse(_host(in1)), mode)
        return core.ascontiguousarray(_reverse(out))
    return _convolve_nd(a, _reverse(_host(b)), mode)


def _check_2d(*arrays):
    for a in arrays:
        if a.ndim != 2:
            raise ValueError("convolve2d inputs must both be 2-D arrays")


def convolve2d(in1, in2, mode="full", boundary="fill", fillvalue=0):
    """2-D convolution with a boundary rule (signaltools.py `convolve2d`)."""
    a, b = _dev(in1), _dev(in2) if isinstance(in2, core.ndarray) else np.asarray(in2)
    _check_2d(a, b)
    if _inputs_swap_needed(mode, a.shape, b.shape):
        a, b = _dev(in2), in1
    return _convolve_nd(a, _host(b), mode, boundary, fillvalue)


def correlate2d(in1, in2, mode="full", boundary="fill", fillvalue=0):
    """2-D cross-correlation with a boundary rule (signaltools.py `correlate2d`)."""
    a, b = _dev(in1), _dev(in2) if isinstance(in2, core.ndarray) else np.asarray(in2)
    _check_2d(a, b)
    if _inputs_swap_needed(mode, a.shape, b.shape):
        out = _convolve_nd(_dev(in2), _reverse(_host(in1)), mode, boundary, fillvalue)
        return core.ascontiguousarray(_reverse(out))
    return _convolve_nd(a, _reverse(_host(b)), mode, boundary, fillvalue, corr2d=True)
