"""Utility that is in the reference but not in SciPy / scikit-image:
n-D convolution as separable convolve1d passes (cupyimg/_misc.py:39-77)."""
import numpy as np

from . import _lib, core
from .scipy.ndimage import convolve1d

__all__ = ["convolve_separable"]


def convolve_separable(x, w, axes=None, **kwargs):
    """Apply the 1-D filter `w` (or one filter per axis) along `axes` of `x`
    (default: every axis); keyword arguments go to `convolve1d`."""
    x = x if isinstance(x, core.ndarray) else core.asarray(np.asarray(x))
    ndim = x.ndim
    axes = tuple(range(ndim)) if axes is None else tuple(axes)
    if any(ax < -ndim or ax > ndim - 1 for ax in axes):
        raise ValueError("axis out of range")
    if isinstance(w, (core.ndarray, np.ndarray)):
        w = [w] * len(axes)
    elif len(w) != len(axes):
        raise ValueError("user should supply one filter per axis")
    for w0 in w:
        if not isinstance(w0, (core.ndarray, np.ndarray)) or w0.ndim != 1:
            raise ValueError("w must be a 1d array (or sequence of 1d arrays)")
    fused = _fused_separable(x, w, axes, kwargs)
    if fused is not None:
        return fused
    for ax, w0 in zip(axes, w):
        x = convolve1d(x, w0, axis=ax, **kwargs)
    return x


def _fused_separable(x, w, axes, kwargs):
    """float32 volumes, odd kernels, default accumulation (`dtype_mode="float"`): all passes in ONE launch of the
    fused separable kernels (mi_separable3d_f32) instead of one generic pass per axis; None when not covered."""
    from .scipy.ndimage import filters as F
    if x.ndim != 3 or x.dtype != np.float32 or set(kwargs) - {"mode", "cval", "origin", "dtype_mode"}:
        return None
    if kwargs.get("dtype_mode", "float") != "float" or not isinstance(kwargs.get("mode", "reflect"), str):
        return None
    origin = kwargs.get("origin", 0)
    if not isinstance(origin, (int, np.integer)):
        return None
    ax3 = [a % 3 for a in axes]
    if len(set(ax3)) != len(ax3):
        return None
    weights, origins = [None, None, None], [0, 0, 0]
    for a, w0 in zip(ax3, w):
        wh = w0.get() if isinstance(w0, core.ndarray) else np.asarray(w0)
        if wh.size % 2 == 0 or wh.size > 33 or wh.dtype.kind not in "fiu":
            return None
        weights[a] = np.ascontiguousarray(wh[::-1], dtype=np.float64)      # convolution = correlation with the flipped kernel
        origins[a] = -int(origin)                                         # ... and the mirrored origin (odd lengths)
    mode = kwargs.get("mode", "reflect")
    out = core.empty(x.shape, np.float32)
    try:
        res = F._fused_3d(x, out, weights, origins, [mode] * 3, float(kwargs.get("cval", 0.0)), False, None)
    except (_lib.Unsupported, ValueError):      # not covered / refused by argument validation: the per-axis passes run
        return None
    return res
