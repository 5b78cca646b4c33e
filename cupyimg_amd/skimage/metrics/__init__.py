"""skimage.metrics subset: the structural similarity index, the end-to-end consumer
of uniform_filter / gaussian_filter (cupyimg/skimage/metrics/_structural_similarity.py:13-251),
and the simple metrics (simple_metrics.py).

Five filtered moments (the fused separable kernel for float32 volumes), one pass that
turns them into the SSIM map (`mi_ssim_combine`) and a strided double-precision sum over
the cropped interior (`mi_sum`); nothing leaves the device except the scalar."""
import ctypes
import warnings

import numpy as np

from ... import core
from ...scipy.ndimage import _support as S
from ...scipy.ndimage.filters import uniform_filter, gaussian_filter

__all__ = ["structural_similarity", "mean_squared_error", "normalized_root_mse", "peak_signal_noise_ratio"]


def _dtype_range(dtype):
    """intensity limits of an image dtype (skimage/util/dtype.py:21-43)"""
    dtype = np.dtype(dtype)
    if dtype.kind == "b":
        return 0, 1
    if dtype.kind in "iu":
        info = np.iinfo(dtype)
        return info.min, info.max
    return -1, 1


def _dev(image):
    return image if isinstance(image, core.ndarray) else core.asarray(np.asarray(image))


def _check_shape_equality(im1, im2):
    if tuple(im1.shape) != tuple(im2.shape):
        raise ValueError("Input images must have the same dimensions.")


def _sum(op, a, b=None):
    out = ctypes.c_double()
    da = a._desc()
    if b is None:
        S.check(S.lib().mi_sum(op, ctypes.byref(da), None, ctypes.byref(out), None))
    else:
        db = b._desc()
        S.check(S.lib().mi_sum(op, ctypes.byref(da), ctypes.byref(db), ctypes.byref(out), None))
    return out.value


def _mul(a, b):
    return S.elementwise("multiply", a, b, core.empty(a.shape, a.dtype))


def _as_float(image, dtype):
    image = _dev(image)
    return core.ascontiguousarray(image if image.dtype == dtype else image.astype(dtype))


def structural_similarity(im1, im2, *, win_size=None, gradient=False, data_range=None, multichannel=False,
                          gaussian_weights=False, full=False, data_dtype=np.float64, **kwargs):
    """Mean structural similarity index between two images
    (_structural_similarity.py:13-251).  Returns the mean as a Python float, plus the
    gradient with respect to im2 and / or the full SSIM image as device arrays."""
    im1, im2 = _dev(im1), _dev(im2)
    _check_shape_equality(im1, im2)
    data_dtype = np.dtype(data_dtype)

    if multichannel:
        args = dict(win_size=win_size, gradient=gradient, data_range=data_range, multichannel=False,
                    gaussian_weights=gaussian_weights, full=full, data_dtype=data_dtype)
        args.update(kwargs)
        nch = im1.shape[-1]
        mssim = np.empty(nch)
        G = core.empty(im1.shape, np.float64) if gradient else None
        Sfull = core.empty(im1.shape, np.float64) if full else None
        for ch in range(nch):
            res = structural_similarity(im1[..., ch], im2[..., ch], **args)
            if gradient and full:
                mssim[ch], G[..., ch], Sfull[..., ch] = res
            elif gradient:
                mssim[ch], G[..., ch] = res
            elif full:
                mssim[ch], Sfull[..., ch] = res
            else:
                mssim[ch] = res
        mssim = float(mssim.mean())
        if gradient and full:
            return mssim, G, Sfull
        if gradient:
            return mssim, G
        if full:
            return mssim, Sfull
        return mssim

    K1 = kwargs.pop("K1", 0.01)
    K2 = kwargs.pop("K2", 0.03)
    sigma = kwargs.pop("sigma", 1.5)
    if K1 < 0:
        raise ValueError("K1 must be positive")
    if K2 < 0:
        raise ValueError("K2 must be positive")
    if sigma < 0:
        raise ValueError("sigma must be positive")
    use_sample_covariance = kwargs.pop("use_sample_covariance", True)

    truncate = 3.5      # an 11-tap filter at the default sigma of 1.5 (Wang et al. 2004)
    if win_size is None:
        win_size = 2 * int(truncate * sigma + 0.5) + 1 if gaussian_weights else 7
    if any(s < win_size for s in im1.shape):
        raise ValueError("win_size exceeds image extent.  If the input is a multichannel (color) image, set "
                         "multichannel=True.")
    if not (win_size % 2 == 1):
        raise ValueError("Window size must be odd.")

    if data_range is None:
        if im1.dtype != im2.dtype:
            warnings.warn("Inputs have mismatched dtype.  Setting data_range based on im1.dtype.", stacklevel=2)
        dmin, dmax = _dtype_range(im1.dtype)
        data_range = dmax - dmin

    ndim = im1.ndim
    if gaussian_weights:
        def filt(a):
            return gaussian_filter(a, sigma=sigma, truncate=truncate, mode="reflect")
    else:
        def filt(a):
            return uniform_filter(a, size=win_size, mode="reflect")

    x = _as_float(im1, data_dtype)
    y = _as_float(im2, data_dtype)
    NP = win_size ** ndim
    cov_norm = NP / (NP - 1) if use_sample_covariance else 1.0

    ux, uy = filt(x), filt(y)
    xx, yy, xy = (core.empty(x.shape, x.dtype) for _ in range(3))
    pd = [a._desc() for a in (x, y, xx, yy, xy)]
    S.check(S.lib().mi_ssim_products(*[ctypes.byref(d) for d in pd], None))       # one read of the pair, three writes
    uxx, uyy, uxy = filt(xx), filt(yy), filt(xy)
    del xx, yy, xy
    C1 = (K1 * data_range) ** 2
    C2 = (K2 * data_range) ** 2

    pad = (win_size - 1) // 2
    if not gradient and 1 <= x.ndim <= 3:
        # the map (only when asked for) and its cropped mean in one pass
        Smap = core.empty(x.shape, data_dtype) if full else None
        descs = [a._desc() for a in (ux, uy, uxx, uyy, uxy)]
        sd = Smap._desc() if full else None
        total = ctypes.c_double(0.0)
        S.check(S.lib().mi_ssim_combine_mean(*[ctypes.byref(d) for d in descs], ctypes.byref(sd) if full else None, pad,
                                             float(cov_norm), float(C1), float(C2), ctypes.byref(total), None))
        count = 1
        for n_ in x.shape:
            count *= n_ - 2 * pad
        mssim = total.value / count
        return (mssim, Smap) if full else mssim

    Smap = core.empty(x.shape, data_dtype)
    fields = [core.empty(x.shape, data_dtype) for _ in range(3)] if gradient else [None] * 3
    descs = [a._desc() for a in (ux, uy, uxx, uyy, uxy, Smap)]
    gdescs = [f._desc() if f is not None else None for f in fields]
    S.check(S.lib().mi_ssim_combine(*[ctypes.byref(d) for d in descs],
                                    *[ctypes.byref(d) if d is not None else None for d in gdescs],
                                    float(cov_norm), float(C1), float(C2), None))

    # ignore a filter-radius strip around the edges
    pad = (win_size - 1) // 2
    inner = Smap[tuple(slice(pad, s - pad) for s in Smap.shape)] if pad else Smap
    mssim = _sum(0, inner) / inner.size

    if gradient:
        # eqs. 7-8 of Avanaki 2009
        grad = _mul(filt(fields[0]), x)
        S.elementwise("add", grad, _mul(filt(fields[1]), y), grad)
        S.elementwise("add", grad, filt(fields[2]), grad)
        grad = S.scale_shift(grad, 2.0 / x.size, 0.0)
        return (mssim, grad, Smap) if full else (mssim, grad)
    return (mssim, Smap) if full else mssim


def _as_floats(im0, im1):
    """both images as one float type (simple_metrics.py:18-23)"""
    im0, im1 = _dev(im0), _dev(im1)
    float_type = np.result_type(im0.dtype, im1.dtype, np.float32)
    return _as_float(im0, float_type), _as_float(im1, float_type)


def mean_squared_error(image0, image1):
    """mean of the squared differences, accumulated in float64 (simple_metrics.py:26-48)"""
    _check_shape_equality(_dev(image0), _dev(image1))
    a, b = _as_floats(image0, image1)
    return _sum(1, a, b) / a.size


def normalized_root_mse(image_true, image_test, *, normalization="euclidean"):
    """NRMSE with the 'euclidean', 'min-max' or 'mean' denominators (simple_metrics.py:51-110)"""
    _check_shape_equality(_dev(image_true), _dev(image_test))
    a, b = _as_floats(image_true, image_test)
    normalization = normalization.lower()
    if normalization == "euclidean":
        denom = np.sqrt(_sum(2, a) / a.size)
    elif normalization == "min-max":
        lo, hi = S.min_max(a)
        denom = hi - lo
    elif normalization == "mean":
        denom = _sum(0, a) / a.size
    else:
        raise ValueError("Unsupported norm_type")
    return float(np.sqrt(mean_squared_error(a, b)) / denom)


def peak_signal_noise_ratio(image_true, image_test, *, data_range=None):
    """PSNR in dB (simple_metrics.py:113-163)"""
    image_true, image_test = _dev(image_true), _dev(image_test)
    _check_shape_equality(image_true, image_test)
    if data_range is None:
        if image_true.dtype != image_test.dtype:
            warnings.warn("Inputs have mismatched dtype.  Setting data_range based on im_true.", stacklevel=2)
        dmin, dmax = _dtype_range(image_true.dtype)
        true_min, true_max = S.min_max(image_true)
        if true_max > dmax or true_min < dmin:
            raise ValueError("im_true has intensity values outside the range expected for its data type.  Please "
                             "manually specify the data_range")
        data_range = dmax - dmin if true_min < 0 else dmax
    err = mean_squared_error(image_true, image_test)
    return float(10 * np.log10((data_range ** 2) / err))
