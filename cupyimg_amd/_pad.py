"""numpy.pad for device arrays (modes constant / edge / wrap / symmetric / reflect): an
order-0 resampling on the larger grid, i.e. one gather kernel whose boundary rule is the
padding rule.  Values travel through double, so 64-bit integers beyond 2**53 are not exact."""
import numpy as np

from . import core

_NDI_MODE = {"constant": "constant", "edge": "nearest", "wrap": "grid-wrap", "symmetric": "reflect", "reflect": "mirror"}


def pad(array, pad_width, mode="constant", constant_values=0):
    from .scipy.ndimage.interpolation import affine_transform
    if mode not in _NDI_MODE:
        raise ValueError("unsupported padding mode '{}'".format(mode))
    if np.isscalar(pad_width):
        pad_width = [(int(pad_width), int(pad_width))] * array.ndim
    pad_width = [(int(p), int(p)) if np.isscalar(p) else (int(p[0]), int(p[1])) for p in pad_width]
    if len(pad_width) != array.ndim:
        raise ValueError("pad_width needs one (before, after) pair per axis")
    if not any(b or a for b, a in pad_width):
        return array.copy()
    shape = tuple(n + b + a for n, (b, a) in zip(array.shape, pad_width))
    src = array.astype(np.uint8) if array.dtype == np.bool_ else array
    out = affine_transform(src, np.eye(array.ndim), offset=[-float(b) for b, _ in pad_width], output_shape=shape, order=0,
                           mode=_NDI_MODE[mode], cval=constant_values, prefilter=False)
    return out.astype(np.bool_) if array.dtype == np.bool_ else out
