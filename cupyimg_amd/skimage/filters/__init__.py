"""skimage.filters subset: gaussian (cupyimg/skimage/filters/_gaussian.py:13-145)."""
import warnings
from collections.abc import Iterable

import numpy as np

from ... import core
from ...scipy import ndimage as ndi
from ...scipy.ndimage import _support as S

__all__ = ["gaussian"]

_INT_RANGE = {np.dtype(t): (np.iinfo(t).min, np.iinfo(t).max)
              for t in (np.uint8, np.uint16, np.uint32, np.int8, np.int16, np.int32)}


def _guess_spatial_dimensions(image):
    """_gaussian.py:148-172: None when (M, N, 3) is ambiguous."""
    if image.ndim == 2:
        return 2
    if image.ndim == 3 and image.shape[-1] != 3:
        return 3
    if image.ndim == 3 and image.shape[-1] == 3:
        return None
    if image.ndim == 4 and image.shape[-1] == 3:
        return 3
    raise ValueError("Expected 2D, 3D, or 4D array, got %iD." % image.ndim)


def _img_as_float(image):
    """skimage.img_as_float for the dtypes the engine carries: floats pass
    through, bool -> {0, 1}, unsigned ints scale to [0, 1], signed to [-1, 1]."""
    dt = image.dtype
    if dt.kind == "f":
        return image
    if dt == np.bool_:
        return image.astype(np.float64)
    if dt not in _INT_RANGE:
        raise ValueError("cannot convert {} images to float".format(dt))
    lo, hi = _INT_RANGE[dt]
    if dt.itemsize <= 2:
        # conversion and scaling in one pass (same arithmetic: the sample as a double, one multiply-add)
        if dt.kind == "u":
            return S.scale_shift(image, 1.0 / hi, 0.0, dtype=np.float64)
        return S.scale_shift(image, 2.0 / (hi - lo), 1.0 / (hi - lo), dtype=np.float64)
    out = image.astype(np.float64)
    if dt.kind == "u":
        return S.scale_shift(out, 1.0 / hi, 0.0)
    return S.scale_shift(out, 2.0 / (hi - lo), 1.0 / (hi - lo))      # (2 x + 1) / (hi - lo)


def convert_to_float(image, preserve_range):
    """_shared/utils.py:393-422"""
    if preserve_range:
        if image.dtype.char not in "df":
            image = image.astype(np.float64)
        return image
    return _img_as_float(image)


def gaussian(image, sigma=1, output=None, mode="nearest", cval=0, multichannel=None, preserve_range=False,
             truncate=4.0):
    """Multi-dimensional Gaussian filter; default mode 'nearest', integer images
    are converted to float, the channel axis (if any) gets sigma 0."""
    image = image if isinstance(image, core.ndarray) else core.asarray(np.asarray(image))
    try:
        spatial_dims = _guess_spatial_dimensions(image)
    except ValueError:
        spatial_dims = image.ndim
    if spatial_dims is None and multichannel is None:
        warnings.warn(RuntimeWarning("Images with dimensions (M, N, 3) are interpreted as 2D+RGB by default. "
                                     "Use `multichannel=False` to interpret as 3D image with last dimension "
                                     "of length 3."))
        multichannel = True
    if not isinstance(sigma, Iterable):
        if sigma < 0:
            raise ValueError("Sigma values less than zero are not valid")
    elif any(s < 0 for s in sigma):
        raise ValueError("Sigma values less than zero are not valid")
    if multichannel:
        if not isinstance(sigma, Iterable):
            sigma = [sigma] * (image.ndim - 1)
        if len(sigma) != image.ndim:
            sigma = tuple(sigma) + (0,)
        sigma = tuple(sigma)
    image = convert_to_float(image, preserve_range)
    if output is None:
        output = core.empty_like(image)
    elif not np.issubdtype(output.dtype, np.floating):
        raise ValueError("Provided output data type is not float")
    ndi.gaussian_filter(image, sigma, output=output, mode=mode, cval=cval, truncate=truncate)
    return output
