"""Utility that is in the reference but not in SciPy / scikit-image:
n-D convolution as separable convolve1d passes (cupyimg/_misc.py:39-77)."""
import numpy as np

from . import core
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
    for ax, w0 in zip(axes, w):
        if not isinstance(w0, (core.ndarray, np.ndarray)) or w0.ndim != 1:
            raise ValueError("w must be a 1d array (or sequence of 1d arrays)")
        x = convolve1d(x, w0, axis=ax, **kwargs)
    return x
