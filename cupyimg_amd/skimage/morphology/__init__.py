"""skimage.morphology subset: erosion, dilation, binary_erosion, binary_dilation.

Behaviour follows cupyimg/skimage/morphology/grey.py:140-262 (+ `_shift_selem`
:21-56, `_invert_selem` :59-89), binary.py:11-78 and misc.py:24-47
(`default_selem`: cross-shaped connectivity-1 element of the image's rank)."""
import functools

import numpy as np

from ... import core
from ...scipy import ndimage as ndi

__all__ = ["erosion", "dilation", "binary_erosion", "binary_dilation"]


def _default_selem(ndim):
    return ndi.generate_binary_structure(ndim, 1)


def default_selem(func):
    """Use a connectivity-1 element of the image's rank when selem is None (misc.py:24-47)."""
    @functools.wraps(func)
    def func_out(image, selem=None, *args, **kwargs):
        if selem is None:
            selem = _default_selem(image.ndim if hasattr(image, "ndim") else np.ndim(image))
        return func(image, selem=selem, *args, **kwargs)
    return func_out


def _host(selem):
    return selem.get() if isinstance(selem, core.ndarray) else np.asarray(selem)


def _shift_selem(selem, shift_x, shift_y):
    """Pad an even-sided 2-D element with one zero row / column so that it has a
    centre; which side is padded moves the element (grey.py:21-56)."""
    if selem.ndim != 2:
        return selem
    m, n = selem.shape
    if m % 2 == 0:
        extra = np.zeros((1, n), selem.dtype)
        selem = np.vstack((selem, extra)) if shift_x else np.vstack((extra, selem))
        m += 1
    if n % 2 == 0:
        extra = np.zeros((m, 1), selem.dtype)
        selem = np.hstack((selem, extra)) if shift_y else np.hstack((extra, selem))
    return selem


def _invert_selem(selem):
    """ndimage.grey_dilation mirrors its footprint; mirror it back (grey.py:59-89)."""
    return selem[(slice(None, None, -1),) * selem.ndim]


def _as_device(image):
    return image if isinstance(image, core.ndarray) else core.asarray(np.asarray(image))


@default_selem
def erosion(image, selem=None, out=None, shift_x=False, shift_y=False):
    """Greyscale erosion: minimum over the neighbourhood (grey.py:140-196)."""
    image = _as_device(image)
    selem = _shift_selem(_host(selem), shift_x, shift_y)
    if out is None:
        out = core.empty_like(image)
    ndi.grey_erosion(image, footprint=selem, output=out)
    return out


@default_selem
def dilation(image, selem=None, out=None, shift_x=False, shift_y=False):
    """Greyscale dilation: maximum over the neighbourhood (grey.py:199-262)."""
    image = _as_device(image)
    selem = _invert_selem(_shift_selem(_host(selem), shift_x, shift_y))
    if out is None:
        out = core.empty_like(image)
    ndi.grey_dilation(image, footprint=selem, output=out)
    return out


@default_selem
def binary_erosion(image, selem=None, out=None):
    """Binary erosion with the border treated as foreground (binary.py:11-44)."""
    image = _as_device(image)
    if out is None:
        out = core.empty(image.shape, np.bool_)
    ndi.binary_erosion(image, structure=_host(selem), output=out, border_value=True)
    return out


@default_selem
def binary_dilation(image, selem=None, out=None):
    """Binary dilation (binary.py:47-78)."""
    image = _as_device(image)
    if out is None:
        out = core.empty(image.shape, np.bool_)
    ndi.binary_dilation(image, structure=_host(selem), output=out)
    return out
