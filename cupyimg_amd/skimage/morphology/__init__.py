"""skimage.morphology subset: grey and binary erosion / dilation / opening / closing,
the top-hats and the structuring-element generators.

Behaviour follows cupyimg/skimage/morphology/grey.py:91-520 (+ `_shift_selem`
:21-56, `_invert_selem` :59-89, `pad_for_eccentric_selems` :91-137), binary.py:11-135,
selem.py and misc.py:24-47 (`default_selem`: cross-shaped connectivity-1 element of
the image's rank)."""
import functools

import numpy as np

from ... import core, _pad
from ...scipy import ndimage as ndi

from ...scipy.ndimage import _support as S

__all__ = ["erosion", "dilation", "opening", "closing", "white_tophat", "black_tophat", "binary_erosion",
           "binary_dilation", "binary_opening", "binary_closing", "square", "rectangle", "diamond", "disk", "cube",
           "octahedron", "ball", "octagon", "star"]


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
    return core.host_copy(selem) if isinstance(selem, core.ndarray) else np.asarray(selem)


def _upload(host):
    """Device array that remembers its host source (core.with_host_hint): erosion(image, disk(3)) does not fetch
    the element back from the device on every call."""
    return core.with_host_hint(core.asarray(host), host)


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


def _pad_edge(image, widths):
    """numpy.pad(image, widths, mode="edge") on the device"""
    return _pad.pad(image, [(w, w) for w in widths], mode="edge")


def pad_for_eccentric_selems(func):
    """Opening / closing with even-sided elements: pad the image by (side - 1) edge
    samples, run, crop (grey.py:91-137)."""
    @functools.wraps(func)
    def func_out(image, selem, out=None, *args, **kwargs):
        image = _as_device(image)
        selem = _host(selem)
        widths = [n - 1 if n % 2 == 0 else 0 for n in selem.shape]
        if not any(widths):
            return func(image, selem, out=out, *args, **kwargs)
        if out is None:
            out = core.empty_like(image)
        res = func(_pad_edge(image, widths), selem, out=None, *args, **kwargs)
        out[...] = res[tuple(slice(w, n - w) for w, n in zip(widths, res.shape))]
        return out
    return func_out


@default_selem
@pad_for_eccentric_selems
def opening(image, selem=None, out=None):
    """Erosion followed by dilation (grey.py:264-311)."""
    eroded = erosion(image, selem)
    return dilation(eroded, selem, out=out, shift_x=True, shift_y=True)


@default_selem
@pad_for_eccentric_selems
def closing(image, selem=None, out=None):
    """Dilation followed by erosion (grey.py:314-361)."""
    dilated = dilation(image, selem)
    return erosion(dilated, selem, out=out, shift_x=True, shift_y=True)


def _subtract_into(out, a, b):
    """out = a - b (xor for boolean images), on the device"""
    if out.dtype == np.bool_:
        out8 = core.empty(out.shape, np.uint8)
        S.elementwise("subtract", a.astype(np.uint8), b.astype(np.uint8), out8)
        out[...] = out8.astype(np.bool_)      # 1 - 0 / 0 - 1 (wraps to 255) are both "different"
    else:
        S.elementwise("subtract", a, b, out)
    return out


@default_selem
def white_tophat(image, selem=None, out=None):
    """Image minus its opening (grey.py:364-437)."""
    image = _as_device(image)
    opened = opening(image, selem)
    if out is None:
        out = core.empty_like(image)
    return _subtract_into(out, image, opened)


@default_selem
def black_tophat(image, selem=None, out=None):
    """Closing minus the image (grey.py:440-513)."""
    image = _as_device(image)
    closed = closing(image, selem)
    if out is None:
        out = core.empty_like(image)
    return _subtract_into(out, closed, image)


@default_selem
def binary_opening(image, selem=None, out=None):
    """Binary erosion then dilation (binary.py:81-108)."""
    eroded = binary_erosion(image, selem)
    return binary_dilation(eroded, selem, out=out)


@default_selem
def binary_closing(image, selem=None, out=None):
    """Binary dilation then erosion (binary.py:111-138)."""
    dilated = binary_dilation(image, selem)
    return binary_erosion(dilated, selem, out=out)


# ---------------------------------------------------------------- structuring elements (selem.py)
# small host-side masks; `_host_*` build them with NumPy, the public functions upload them
def _host_ball_like(radius, ndim, norm, dtype):
    n = int(2 * radius + 1)
    axes = np.meshgrid(*([np.linspace(-radius, radius, n)] * ndim), indexing="ij", sparse=True)
    if norm == 1:
        return (sum(np.abs(a) for a in axes) <= radius).astype(dtype)
    return (sum(a * a for a in axes) <= radius * radius).astype(dtype)


def _host_octagon(m, n, dtype=np.uint8):
    """the square of side m + 2n with its four corners cut along the diagonals
    (= the convex hull of the eight vertices selem.py marks)"""
    side = m + 2 * n
    i, j = np.ogrid[:side, :side]
    keep = (i + j >= n) & (i + j <= 2 * (side - 1) - n) & (i - j <= m + n - 1) & (j - i <= m + n - 1)
    return keep.astype(dtype)


def _host_star(a, dtype=np.uint8):
    """square of side 2a + 1 overlaid with the diamond through the midpoints of an
    a // 2 wider frame (8 vertices)"""
    if a == 1:
        return np.ones((3, 3), dtype)
    m, n = 2 * a + 1, a // 2
    side = m + 2 * n
    c = (side - 1) // 2
    i, j = np.ogrid[:side, :side]
    sq = (i >= n) & (i < m + n) & (j >= n) & (j < m + n)
    rot = np.abs(i - c) + np.abs(j - c) <= c
    return (sq | rot).astype(dtype)


def square(width, dtype=np.uint8):
    return _upload(np.ones((width, width), dtype=dtype))


def rectangle(nrows, ncols, dtype=np.uint8):
    return _upload(np.ones((nrows, ncols), dtype=dtype))


def cube(width, dtype=np.uint8):
    return _upload(np.ones((width, width, width), dtype=dtype))


def diamond(radius, dtype=np.uint8):
    """|i| + |j| <= radius on a (2 radius + 1)^2 grid"""
    return _upload(_host_ball_like(int(radius), 2, 1, dtype))


def disk(radius, dtype=np.uint8):
    """i^2 + j^2 <= radius^2"""
    return _upload(_host_ball_like(int(radius), 2, 2, dtype))


def octahedron(radius, dtype=np.uint8):
    """3-D |.|_1 ball; non-integer radii allowed as in the reference"""
    return _upload(_host_ball_like(radius, 3, 1, dtype))


def ball(radius, dtype=np.uint8):
    return _upload(_host_ball_like(radius, 3, 2, dtype))


def octagon(m, n, dtype=np.uint8):
    return _upload(_host_octagon(m, n, dtype))


def star(a, dtype=np.uint8):
    return _upload(_host_star(a, dtype))
