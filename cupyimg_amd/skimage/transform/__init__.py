"""skimage.transform subset: warp on top of ndimage.map_coordinates
(cupyimg/skimage/transform/_warps.py:790-1028; mode translation :163-169,
`warp_coords` :640-742, output clipping :745-787).

`inverse_map` may be a coordinate array of shape (ndim, *output_shape), a 3x3
homogeneous matrix acting on (x, y) column/row coordinates, or a callable
mapping (N, 2) output (x, y) pairs to input (x, y) pairs.  Spline orders 0-5
(prefilter for orders > 1; the default for non-bool images is 1, as in the reference)."""
import numpy as np

from ... import core
from ...scipy import ndimage as ndi
from ...scipy.ndimage import _support as S
from ..filters import convert_to_float

__all__ = ["warp", "warp_coords"]

_NDI_MODE = {"constant": "constant", "edge": "nearest", "symmetric": "reflect", "reflect": "mirror",
             "wrap": "wrap"}


def _to_ndimage_mode(mode):
    """numpy.pad names -> ndimage names (_warps.py:163-169, _geometric.py:14-21)."""
    if mode not in _NDI_MODE:
        raise ValueError("Unknown mode: '{}', or cannot translate mode. The mode should be one of "
                         "'constant', 'edge', 'symmetric', 'reflect', or 'wrap'.".format(mode))
    return _NDI_MODE[mode]


def _validate_interpolation_order(image_dtype, order):
    """_shared/utils.py:425-464"""
    if order is None:
        return 0 if image_dtype == np.bool_ else 1
    if order < 0 or order > 5:
        raise ValueError("Spline interpolation order has to be in the range 0-5.")
    return order


def warp_coords(coord_map, shape, dtype=np.float64):
    """Source coordinates for every output pixel of `shape` (rows, cols[, bands]);
    `coord_map` works on (N, 2) arrays of (col, row) pairs (_warps.py:640-742)."""
    shape = tuple(int(s) for s in shape)
    rows, cols = shape[0], shape[1]
    coords_shape = [len(shape), rows, cols] + ([shape[2]] if len(shape) == 3 else [])
    coords = np.empty(coords_shape, dtype=dtype)
    tf = np.indices((cols, rows), dtype=dtype).reshape(2, -1).T
    tf = np.asarray(coord_map(tf))
    tf = tf.T.reshape((-1, cols, rows)).swapaxes(1, 2)
    if len(shape) == 3:
        coords[1, ...] = tf[0][..., None]
        coords[0, ...] = tf[1][..., None]
        coords[2, ...] = np.arange(shape[2], dtype=dtype)
    else:
        coords[1, ...] = tf[0]
        coords[0, ...] = tf[1]
    return coords


def warp(image, inverse_map, map_args={}, output_shape=None, order=None, mode="constant", cval=0.0, clip=True,
         preserve_range=False):
    """Warp an image according to an inverse coordinate map (_warps.py:790-1028)."""
    image = image if isinstance(image, core.ndarray) else core.asarray(np.asarray(image))
    if image.size == 0:
        raise ValueError("Cannot warp empty image with dimensions", image.shape)
    order = _validate_interpolation_order(image.dtype, order)
    image = convert_to_float(image, preserve_range)
    input_shape = tuple(image.shape)
    output_shape = input_shape if output_shape is None else tuple(int(s) for s in output_shape)

    if isinstance(inverse_map, core.ndarray):
        coords = inverse_map
    else:
        if isinstance(inverse_map, np.ndarray) and inverse_map.shape == (3, 3):
            H = np.asarray(inverse_map, dtype=np.float64)

            def inverse_map(xy, H=H):
                src = np.c_[xy, np.ones(len(xy))] @ H.T
                return src[:, :2] / src[:, 2:3]
        if isinstance(inverse_map, np.ndarray):
            coords = inverse_map
        else:
            if image.ndim < 2 or image.ndim > 3:
                raise ValueError("Only 2-D images (grayscale or color) are supported, when providing a "
                                 "callable `inverse_map`.")
            if len(input_shape) == 3 and len(output_shape) == 2:
                output_shape = (output_shape[0], output_shape[1], input_shape[2])
            coords = warp_coords(lambda xy: inverse_map(xy, **map_args), output_shape)
        coords = core.asarray(np.ascontiguousarray(coords))

    warped = ndi.map_coordinates(image, coords, prefilter=order > 1, mode=_to_ndimage_mode(mode), order=order,
                                 cval=cval)
    if clip and order != 0:
        # clip to the input range, keeping cval where it marks the outside (_warps.py:745-787)
        lo, hi = S.min_max(image)
        keep = (mode == "constant") and not (lo <= cval <= hi)
        warped = S.clip(warped, lo, hi, cval if keep else None)
    return warped
