"""HIP path vs the CPU oracle on the same seeded inputs, at sizes the oracle
finishes in seconds, plus API-level behaviour the reference's tests cover
(output argument forms, aliasing, strided views, thread safety, errors)."""
import threading

import numpy as np
import pytest

from _cases import maxnorm_rel
from oracle import ndimage as orc

pytestmark = pytest.mark.gpu
MODES = ["reflect", "constant", "nearest", "mirror", "wrap"]


@pytest.fixture(scope="module")
def ndi(gpu):
    from cupyimg_amd.scipy import ndimage
    return ndimage


# ------------------------------------------------------------------ fused f32
@pytest.mark.parametrize("shape", [(40, 37, 64), (19, 50, 256), (33, 21, 264), (70, 40, 512), (9, 5, 8)])
@pytest.mark.parametrize("size", [3, 5, 7, 9])
def test_uniform_filter_fused_f32(gpu, ndi, shape, size):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape).astype(np.float32)
    xd = gpu.asarray(x)
    for mode in MODES:
        ref = orc.uniform_filter(x, size, mode=mode, cval=0.75)
        got = ndi.uniform_filter(xd, size, mode=mode, cval=0.75).get()
        r = maxnorm_rel(got, ref)
        assert r <= 1e-6, (shape, size, mode, r)


def test_uniform_filter_fused_mixed_axes(gpu, ndi):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((30, 45, 128)).astype(np.float32)
    xd = gpu.asarray(x)
    for size, mode, origin in [((5, 1, 3), "reflect", 0), ((1, 7, 1), "mirror", 0), ((3, 5, 9), ["wrap", "nearest", "reflect"], (1, -1, 0)),
                               ((9, 9, 1), "constant", 0), ((1, 1, 5), "nearest", 0)]:
        ref = orc.uniform_filter(x, size, mode=mode, origin=origin, cval=-0.5)
        got = ndi.uniform_filter(xd, size, mode=mode, origin=origin, cval=-0.5).get()
        assert maxnorm_rel(got, ref) <= 1e-6, (size, mode, origin)


@pytest.mark.parametrize("sigma", [0.5, 1.0, (1.0, 0.6, 0.8)])
def test_gaussian_filter_fused_f32(gpu, ndi, sigma):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((36, 41, 96)).astype(np.float32)
    xd = gpu.asarray(x)
    for mode in MODES:
        for order in [0, 1]:
            ref = orc.gaussian_filter(x, sigma, order=order, mode=mode, cval=0.0)
            got = ndi.gaussian_filter(xd, sigma, order=order, mode=mode, cval=0.0).get()
            assert maxnorm_rel(got, ref) <= 1e-6, (sigma, mode, order)


@pytest.mark.parametrize("shape", [(40, 40, 40), (37, 29, 264), (20, 70, 512)])
def test_gaussian_long_kernels_streaming(gpu, ndi, shape):
    """> 9 taps per axis: two streaming launches (x fused into the z pass) -- stream3d.hip"""
    rng = np.random.default_rng(6)
    x = rng.standard_normal(shape).astype(np.float32)
    xd = gpu.asarray(x)
    for sigma, mode in [(2.0, "reflect"), (2.0, "mirror"), (3.0, "nearest"), (2.5, "wrap"), (2.0, "constant"),
                        (4.0, "mirror"), (3.5, "wrap"), (3.0, "constant"),
                        ((2.0, 1.0, 3.0), "reflect"), ((0.0, 2.0, 0.0), "mirror"), ((1.5, 0.0, 4.0), "reflect")]:
        ref = orc.gaussian_filter(x, sigma, mode=mode, cval=0.3)
        got = ndi.gaussian_filter(xd, sigma, mode=mode, cval=0.3).get()
        assert maxnorm_rel(got, ref) <= 1e-6, (shape, sigma, mode, maxnorm_rel(got, ref))
    for size in [11, 15, (13, 5, 17), 21, (3, 3, 33), (25, 1, 29)]:
        ref = orc.uniform_filter(x, size, mode="reflect")
        got = ndi.uniform_filter(xd, size, mode="reflect").get()
        assert maxnorm_rel(got, ref) <= 1e-6, (shape, size, maxnorm_rel(got, ref))


# ------------------------------------------------------------------ generic
@pytest.mark.parametrize("dtype", ["bool", "int8", "uint8", "int16", "uint16", "int32", "uint32", "int64",
                                   "uint64", "float32", "float64"])
def test_dtype_matrix_correlate(gpu, ndi, dtype):
    """every input dtype x every output dtype (reference tests/test_ndimage.py:238-263)"""
    rng = np.random.default_rng(7)
    x = (rng.random((9, 11)) * 20).astype(dtype)
    w = np.array([[1, 0, 2], [0, 1, 0], [1, 0, -1]], dtype=np.float64)
    xd = gpu.asarray(x)
    for odt in ["int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "float32", "float64"]:
        ref = orc.correlate(x, w, output=odt)
        got = ndi.correlate(xd, w, output=np.dtype(odt)).get()
        assert got.dtype == np.dtype(odt)
        assert np.array_equal(got, ref), (dtype, odt)
        ref = orc.correlate1d(x, [1, 2, 1], axis=0, output=odt)
        got = ndi.correlate1d(xd, [1, 2, 1], axis=0, output=np.dtype(odt), dtype_mode="ndimage").get()
        assert np.array_equal(got, ref), (dtype, odt)
        ref = orc.minimum_filter(x, size=3, output=odt)
        got = ndi.minimum_filter(xd, size=3, output=np.dtype(odt)).get()
        assert np.array_equal(got, ref), (dtype, odt)


def test_uniform_integer_exact(gpu, ndi):
    """gh-6930 style: integer box means must truncate exact sums
    (reference xfail: tests/test_filters.py:444-450)."""
    x = np.arange(1, 40, dtype=np.uint8).reshape(3, 13)
    for size in [2, 3, 4, 6]:
        ref = orc.uniform_filter(x, size)
        got = ndi.uniform_filter(gpu.asarray(x), size).get()
        assert np.array_equal(got, ref)


def test_long_kernel_and_short_signal(gpu, ndi):
    rng = np.random.default_rng(8)
    x = rng.standard_normal((5, 3))
    w = rng.standard_normal(131)     # longer than the 64 inline taps -> scratch upload
    for mode in MODES:
        ref = orc.correlate1d(x, w, axis=1, mode=mode)
        got = ndi.correlate1d(gpu.asarray(x), w, axis=1, mode=mode, dtype_mode="ndimage").get()
        assert np.array_equal(got, ref), mode


def test_4d_and_5d(gpu, ndi):
    rng = np.random.default_rng(9)
    x = rng.standard_normal((4, 5, 6, 7)).astype(np.float32)
    xd = gpu.asarray(x)
    assert maxnorm_rel(ndi.uniform_filter(xd, 3).get(), orc.uniform_filter(x, 3)) <= 1e-6
    w = rng.standard_normal((2, 3, 1, 3))
    assert np.array_equal(ndi.correlate(xd, w, mode="mirror").get(), orc.correlate(x, w, mode="mirror"))
    fp = rng.random((3, 2, 3, 2)) > 0.3
    assert np.array_equal(ndi.maximum_filter(xd, footprint=fp).get(), orc.maximum_filter(x, footprint=fp))
    b = x > 0
    assert np.array_equal(ndi.binary_erosion(gpu.asarray(b)).get(), orc.binary_erosion(b))
    c = rng.uniform(-1, 7, size=(4, 50))
    got = ndi.map_coordinates(xd, c, order=1, mode="nearest").get()
    assert np.allclose(got, orc.map_coordinates(x, c, order=1, mode="nearest"), atol=1e-6)
    x5 = rng.integers(0, 9, size=(3, 3, 4, 3, 5)).astype(np.int32)
    got = ndi.map_coordinates(gpu.asarray(x5), rng.uniform(0, 2, size=(5, 20)), order=1, mode="reflect").get()
    assert got.dtype == np.int32


# ------------------------------------------------------------------ API forms
def test_output_forms_and_aliasing(gpu, ndi):
    rng = np.random.default_rng(10)
    x = rng.standard_normal((12, 14, 16)).astype(np.float32)
    ref = orc.uniform_filter(x, 3)
    xd = gpu.asarray(x)
    out = gpu.empty(x.shape, np.float32)
    r = ndi.uniform_filter(xd, 3, output=out)
    assert r is out and maxnorm_rel(out.get(), ref) <= 1e-6
    # output dtype
    r = ndi.uniform_filter(xd, 3, output=np.float64)
    assert r.dtype == np.float64 and maxnorm_rel(r.get(), orc.uniform_filter(x, 3, output=np.float64)) <= 1e-12
    # in place (output is input): reference handles this with a temp (_filters_core.py:148-155)
    y = gpu.asarray(x)
    ndi.uniform_filter(y, 3, output=y)
    assert maxnorm_rel(y.get(), ref) <= 1e-6
    y = gpu.asarray(x)
    ndi.correlate1d(y, [1.0, 2.0, 3.0], axis=1, output=y, dtype_mode="ndimage")
    assert np.array_equal(y.get(), orc.correlate1d(x, [1.0, 2.0, 3.0], axis=1))
    # wrong output shape
    with pytest.raises(ValueError):
        ndi.uniform_filter(xd, 3, output=gpu.empty((3, 3, 3), np.float32))


def test_strided_views(gpu, ndi):
    rng = np.random.default_rng(11)
    x = rng.standard_normal((10, 12, 16)).astype(np.float32)
    xd = gpu.asarray(x)
    v = xd[::2, 1:, ::-1]
    xv = x[::2, 1:, ::-1]
    assert np.array_equal(v.get(), xv)
    assert maxnorm_rel(ndi.uniform_filter(v, 3).get(), orc.uniform_filter(xv, 3)) <= 1e-6
    t = xd.transpose(2, 0, 1)
    assert np.array_equal(ndi.minimum_filter(t, size=3).get(), orc.minimum_filter(x.transpose(2, 0, 1), size=3))
    big = gpu.zeros((10, 12, 32), np.float32)
    ov = big[:, :, ::2]
    ndi.gaussian_filter(xd, 1.0, output=ov)
    assert maxnorm_rel(ov.get(), orc.gaussian_filter(x, 1.0)) <= 1e-6


def test_thread_safety(gpu, ndi):
    """Same op from 4 host threads on separate outputs
    (reference tests/test_filters.py:354-412)."""
    rng = np.random.default_rng(12)
    x = rng.standard_normal((20, 24, 32)).astype(np.float32)
    xd = gpu.asarray(x)
    ref = orc.uniform_filter(x, 5)
    refm = orc.maximum_filter(x, size=3)
    outs = [None] * 8

    def work(i):
        if i % 2:
            outs[i] = ndi.uniform_filter(xd, 5)
        else:
            outs[i] = ndi.maximum_filter(xd, size=3)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for i, o in enumerate(outs):
        if i % 2:
            assert maxnorm_rel(o.get(), ref) <= 1e-6
        else:
            assert np.array_equal(o.get(), refm)


def test_error_behaviour(gpu, ndi):
    """Exception types of the reference."""
    x = gpu.asarray(np.zeros((5, 6), np.float32))
    with pytest.raises(RuntimeError):
        ndi.uniform_filter(x, 3, mode="bogus")
    with pytest.raises(RuntimeError):
        ndi.correlate(x, np.ones((3,)))                      # rank mismatch
    with pytest.raises(RuntimeError):
        ndi.uniform_filter1d(x, 0)
    with pytest.raises(RuntimeError):
        ndi.correlate1d(x, np.ones((2, 2)))
    with pytest.raises(RuntimeError):
        ndi.uniform_filter(x, (3, 3, 3))
    with pytest.raises(ValueError):
        ndi.correlate1d(x, [1, 1, 1], origin=2)
    with pytest.raises(ValueError):
        ndi.correlate1d(x, [1, 1, 1], origin=-2)
    with pytest.raises(ValueError):
        ndi.minimum_filter(x, footprint=np.zeros((3, 3), bool))
    with pytest.raises(ValueError):
        ndi.gaussian_filter(x, 1.0, order=-1)
    with pytest.raises(ValueError):
        ndi.map_coordinates(x, np.zeros((2, 3)), order=7)
    with pytest.raises(ValueError):
        ndi.map_coordinates(x, np.zeros((2, 3)), order=1, mode="bogus")
    with pytest.raises(ValueError):
        ndi.map_coordinates(x, np.zeros((2, 3), dtype=np.complex64), order=1)
    with pytest.raises(TypeError):
        ndi.binary_erosion(x, iterations=1.5)
    with pytest.raises(NotImplementedError):
        ndi.minimum_filter(x, size=3, cval=np.nan)
    with pytest.raises(NotImplementedError):
        ndi.uniform_filter(gpu.asarray(np.zeros((4, 4), np.int32)), 3, mode="constant", cval=np.inf)
    with pytest.raises(RuntimeError):
        ndi.binary_erosion(x, structure=np.ones((3,), bool))
    with pytest.raises(ValueError):
        ndi.grey_erosion(x)
    with pytest.raises(RuntimeError):
        ndi.minimum_filter(x)
    # empty input (reference test_correlate09/10)
    e = ndi.correlate(gpu.asarray(np.zeros((0,), np.float64)), np.array([1.0, 1.0]))
    assert e.shape == (0,)
    e = ndi.uniform_filter(gpu.asarray(np.zeros((3, 0), np.float32)), 3)
    assert e.shape == (3, 0)


# ------------------------------------------------------------------ morphology
@pytest.mark.parametrize("shape", [(64, 70, 80), (31, 257)])
def test_grey_morphology_u8(gpu, ndi, shape):
    rng = np.random.default_rng(13)
    x = rng.integers(0, 256, size=shape).astype(np.uint8)
    xd = gpu.asarray(x)
    for size in [3, 7]:
        for mode in MODES:
            assert np.array_equal(ndi.grey_erosion(xd, size=size, mode=mode, cval=9).get(),
                                  orc.grey_erosion(x, size=size, mode=mode, cval=9)), (size, mode)
            assert np.array_equal(ndi.grey_dilation(xd, size=size, mode=mode, cval=9).get(),
                                  orc.grey_dilation(x, size=size, mode=mode, cval=9)), (size, mode)


@pytest.mark.parametrize("shape", [(20, 24, 32), (11, 13, 1040), (6, 9, 2048), (9, 11, 48)])
def test_grey_morphology_u8_streaming(gpu, ndi, shape):
    """uint8 volumes, odd flat sizes: two streaming launches (minmax3d_u8.hip), bit-exact"""
    rng = np.random.default_rng(21)
    x = rng.integers(0, 256, size=shape).astype(np.uint8)
    xd = gpu.asarray(x)
    for size in [3, 5, 7, 9, (3, 7, 5), (1, 9, 3), (7, 1, 1), (1, 1, 9), (11, 13, 3)]:
        for mode in MODES:
            for fn in ["grey_erosion", "grey_dilation", "minimum_filter", "maximum_filter"]:
                ref = getattr(orc, fn)(x, size=size, mode=mode, cval=77)
                got = getattr(ndi, fn)(xd, size=size, mode=mode, cval=77).get()
                assert np.array_equal(got, ref), (shape, size, mode, fn, int((got != ref).sum()))
    # cval that is not a uint8 value takes the generic path and still matches SciPy semantics
    ref = orc.maximum_filter(x, size=3, mode="constant", cval=300)
    assert np.array_equal(ndi.maximum_filter(xd, size=3, mode="constant", cval=300).get(), ref)


def test_binary_iterations(gpu, ndi):
    rng = np.random.default_rng(14)
    x = rng.random((40, 50, 60)) > 0.35
    mask = rng.random(x.shape) > 0.2
    xd, md = gpu.asarray(x), gpu.asarray(mask)
    for it in [1, 2, 5, 0]:
        for fn in ["binary_erosion", "binary_dilation"]:
            ref = getattr(orc, fn)(x, iterations=it, mask=mask)
            got = getattr(ndi, fn)(xd, iterations=it, mask=md).get()
            assert got.dtype == np.bool_ and np.array_equal(got, ref), (fn, it)
    ref = orc.binary_dilation(x, np.ones((3, 3, 3)), iterations=3, border_value=1)
    got = ndi.binary_dilation(xd, np.ones((3, 3, 3)), iterations=3, border_value=1).get()
    assert np.array_equal(got, ref)
    # composites
    import scipy.ndimage as sndi
    assert np.array_equal(ndi.binary_opening(xd).get(), sndi.binary_opening(x))
    assert np.array_equal(ndi.binary_closing(xd, iterations=2).get(), sndi.binary_closing(x, iterations=2))
    assert np.array_equal(ndi.binary_fill_holes(xd).get(), sndi.binary_fill_holes(x))
    assert np.array_equal(ndi.binary_propagation(gpu.asarray(x & mask), mask=md).get(),
                          sndi.binary_propagation(x & mask, mask=mask))
    assert np.array_equal(ndi.binary_hit_or_miss(xd).get(), sndi.binary_hit_or_miss(x))


# ------------------------------------------------------------------ interpolation
def test_affine_and_map_3d(gpu, ndi):
    rng = np.random.default_rng(15)
    n = 48
    x = rng.standard_normal((n, n, n)).astype(np.float32)
    ang = np.deg2rad(7.0)
    R = np.array([[1, 0, 0], [0, np.cos(ang), -np.sin(ang)], [0, np.sin(ang), np.cos(ang)]])
    M = np.diag([1.02, 1.0, 1.0]) @ R
    ctr = (n - 1) / 2.0
    off = ctr - M @ np.array([ctr] * 3) + np.array([0.5, -1.25, 2.0])
    xd = gpu.asarray(x)
    for mode in ["constant", "grid-constant", "nearest", "mirror", "reflect", "wrap", "grid-wrap"]:
        for order in [0, 1]:
            ref = orc.affine_transform(x, M, off, order=order, mode=mode, cval=0.25)
            got = ndi.affine_transform(xd, M, off, order=order, mode=mode, cval=0.25).get()
            # same double arithmetic on both sides; float32 store
            assert np.allclose(got, ref, rtol=0, atol=1e-6), (mode, order, np.abs(got - ref).max())
    idx = np.indices((n, n, n)).reshape(3, -1).astype(np.float64)
    coords = (M @ idx + off[:, None]).reshape(3, n, n, n).astype(np.float32)
    ref = orc.map_coordinates(x, coords, order=1, mode="constant")
    got = ndi.map_coordinates(xd, gpu.asarray(coords), order=1, mode="constant").get()
    assert np.allclose(got, ref, rtol=0, atol=1e-6)
    with pytest.raises(ValueError):
        ndi.map_coordinates(xd, gpu.asarray(coords), order=6)


# ------------------------------------------------------------------ skimage facade
def test_skimage_facade(gpu):
    import scipy.ndimage as sndi

    from cupyimg_amd.skimage import filters as skf
    from cupyimg_amd.skimage import morphology as skm
    from cupyimg_amd.skimage import transform as skt
    rng = np.random.default_rng(30)
    img = rng.integers(0, 256, size=(40, 50)).astype(np.uint8)
    d = gpu.asarray(img)
    # default cross element; reference literal cases live in skimage/morphology/tests/test_grey.py
    cross = sndi.generate_binary_structure(2, 1)
    assert np.array_equal(skm.erosion(d).get(), sndi.grey_erosion(img, footprint=cross))
    assert np.array_equal(skm.dilation(d).get(), sndi.grey_dilation(img, footprint=cross[::-1, ::-1]))
    sq = np.ones((4, 4), np.uint8)                                   # even-sided element: shifted
    ref = sndi.grey_erosion(img, footprint=np.pad(sq, ((1, 0), (1, 0))))
    assert np.array_equal(skm.erosion(d, sq).get(), ref)
    b = img > 128
    assert np.array_equal(skm.binary_erosion(gpu.asarray(b)).get(), sndi.binary_erosion(b, cross, border_value=True))
    assert np.array_equal(skm.binary_dilation(gpu.asarray(b)).get(), sndi.binary_dilation(b, cross))
    # gaussian: default mode 'nearest', uint8 -> float in [0, 1]
    g = skf.gaussian(d, sigma=1.5).get()
    ref = sndi.gaussian_filter(img / 255.0, 1.5, mode="nearest")
    assert g.dtype == np.float64 and np.allclose(g, ref, atol=1e-12)
    f = rng.standard_normal((20, 24, 3)).astype(np.float32)
    g = skf.gaussian(gpu.asarray(f), sigma=2, multichannel=True).get()
    assert np.abs(g - sndi.gaussian_filter(f, (2, 2, 0), mode="nearest")).max() <= 1e-6
    with pytest.raises(ValueError):
        skf.gaussian(d, sigma=-1)
    # warp: callable inverse map and 3x3 matrix, order 1 default, edge mode -> nearest
    fimg = rng.random((30, 36))
    shift = lambda xy: xy + np.array([2.5, -1.25])
    w = skt.warp(gpu.asarray(fimg), shift, mode="edge").get()
    yy, xx = np.indices(fimg.shape, dtype=np.float64)
    ref = sndi.map_coordinates(fimg, [yy - 1.25, xx + 2.5], order=1, mode="nearest")
    assert np.allclose(w, ref, atol=1e-12)
    H = np.array([[1.0, 0.1, 1.0], [-0.05, 0.95, 2.0], [0.0, 0.0, 1.0]])
    w = skt.warp(gpu.asarray(fimg), H, order=1).get()
    src = np.stack([xx.ravel(), yy.ravel(), np.ones(xx.size)]) .T @ H.T
    ref = sndi.map_coordinates(fimg, [src[:, 1].reshape(fimg.shape), src[:, 0].reshape(fimg.shape)], order=1,
                               mode="constant", cval=0.0)
    exp = np.clip(ref, fimg.min(), fimg.max())
    exp[ref == 0.0] = 0.0          # cval outside the input range is preserved (_warps.py:779-787)
    assert np.allclose(w, exp, atol=1e-12)
    # cubic warp (prefilter on the device); clipping keeps the overshoot inside the input range
    w3 = skt.warp(gpu.asarray(fimg), H, order=3).get()
    ref3 = sndi.map_coordinates(fimg, [src[:, 1].reshape(fimg.shape), src[:, 0].reshape(fimg.shape)], order=3,
                                mode="constant", cval=0.0)
    exp3 = np.clip(ref3, fimg.min(), fimg.max())
    exp3[ref3 == 0.0] = 0.0
    assert np.allclose(w3, exp3, atol=1e-10)
    # signed integer images scale to [-1, 1]: (2 x + 1) / (max - min)
    i16 = rng.integers(-3000, 3000, size=(12, 14)).astype(np.int16)
    g = skf.gaussian(gpu.asarray(i16), sigma=1.0).get()
    ref = sndi.gaussian_filter((2.0 * i16.astype(np.float64) + 1.0) / 65535.0, 1.0, mode="nearest")
    assert np.allclose(g, ref, atol=1e-12)


# ------------------------------------------------------------------ > 4 GiB volumes
def test_fused_uniform_filter_beyond_4gib(gpu, ndi):
    """Slabs of the 2048^3 config exceed 32-bit byte offsets: planes are
    addressed through per-plane descriptors.  Checked against the oracle on
    sub-slabs around the 2 GiB / 4 GiB crossings and at both ends (z wrap maps
    the first planes to the far end of the volume)."""
    nz, ny, nx = 1100, 1024, 1024                     # 4.3 GiB per array
    rng = np.random.default_rng(21)
    base = rng.standard_normal((44, ny, nx)).astype(np.float32)
    x = np.empty((nz, ny, nx), np.float32)
    for z0 in range(0, nz, 44):
        n = min(44, nz - z0)
        x[z0:z0 + n] = base[:n] + np.float32(0.01) * np.arange(z0, z0 + n, dtype=np.float32)[:, None, None]
    xd = gpu.asarray(x)
    out = gpu.empty(x.shape, np.float32)
    for size, mode in [(9, "wrap"), (5, "reflect")]:
        r = size // 2
        ndi.uniform_filter(xd, size, mode=mode, output=out)
        for zc in (0, 512, 1024, nz - 3):
            a, b = max(zc - 3, 0), min(zc + 3, nz)
            idx = np.arange(a - r, b + r)
            if mode == "wrap":
                sub = x[idx % nz]
                ref = orc.uniform_filter(sub, size, mode=["nearest", mode, mode])[r:-r]
            else:
                lo_pad, hi_pad = a - r < 0, b + r > nz
                idx = idx[(idx >= 0) & (idx < nz)]
                sub = x[idx]
                ref = orc.uniform_filter(sub, size, mode=mode)
                ref = ref[(0 if lo_pad else r):(len(idx) if hi_pad else len(idx) - r)]
            got = out[a:b].get()
            assert got.shape == ref.shape
            assert maxnorm_rel(got, ref) <= 1e-6, (size, mode, zc)
    del xd, out
    gpu.free_all_blocks()


# ------------------------------------------------------------------ LDS-tiled dense stencil
def _stencil_cases():
    rng = np.random.default_rng(31)
    lap = np.zeros((3, 3, 3)); lap[1, 1, :] = 1; lap[1, :, 1] = 1; lap[:, 1, 1] = 1; lap[1, 1, 1] = -6
    return [
        ((20, 37, 64), rng.standard_normal((3, 3, 3)), 0),
        ((9, 50, 264), rng.standard_normal((5, 3, 5)), 0),
        ((33, 18, 520), rng.standard_normal((3, 5, 7)), (1, -2, 0)),
        ((17, 40, 256), rng.standard_normal((2, 4, 6)), (0, 1, -1)),          # even extents, origins
        ((12, 30, 128), lap, 0),                                             # zero weights are skipped
        ((8, 20, 72), rng.standard_normal((7, 7, 7)), 0),                     # 8-row tiles
        ((6, 7, 8), rng.standard_normal((5, 7, 9)), (0, 0, 0)),               # window nearly as large as the array
        ((40, 300), rng.standard_normal((5, 5)), 0),                         # 2-D image
        ((16, 16, 16), rng.standard_normal((1, 1, 3)), (0, 0, 1)),
    ]


@pytest.mark.parametrize("case", range(9))
def test_tiled_stencil_matches_generic_kernel_and_oracle(gpu, ndi, case):
    """correlate / convolve on float32: the LDS-tiled kernel (stencil3d.hip)
    accumulates the taps in the same order and precision as the generic kernel,
    so the two agree bit for bit; both match the oracle."""
    import ctypes
    from cupyimg_amd import _lib
    lib = _lib.load()
    lib.mi_debug_set_stencil.argtypes = [ctypes.c_int]
    shape, w, origin = _stencil_cases()[case]
    rng = np.random.default_rng(32 + case)
    x = rng.standard_normal(shape).astype(np.float32)
    x[tuple(s // 2 for s in shape)] = np.inf                              # 0 * inf must not appear for skipped taps
    xd = gpu.asarray(x)
    for fn, ofn in [(ndi.correlate, orc.correlate), (ndi.convolve, orc.convolve)]:
        for mode in MODES:
            for dtype_mode in ("ndimage", "float"):
                try:
                    lib.mi_debug_set_stencil(1)
                    tiled = fn(xd, w, mode=mode, cval=0.75, origin=origin, dtype_mode=dtype_mode).get()
                    lib.mi_debug_set_stencil(0)
                    generic = fn(xd, w, mode=mode, cval=0.75, origin=origin, dtype_mode=dtype_mode).get()
                finally:
                    lib.mi_debug_set_stencil(1)
                assert np.array_equal(tiled, generic, equal_nan=True), (fn.__name__, mode, dtype_mode)
            ref = ofn(x, w, mode=mode, cval=0.75, origin=origin)
            fin = np.isfinite(ref)
            assert np.array_equal(fin, np.isfinite(generic))
            assert maxnorm_rel(np.where(fin, generic, 0), np.where(fin, ref, 0)) <= 1e-6, (fn.__name__, mode)


def test_tiled_stencil_large_volume_property(gpu, ndi):
    """Full-size check without a CPU reference: a dense 3x3x3 correlate with a
    separable (outer product) kernel equals three 1-D correlations."""
    rng = np.random.default_rng(40)
    x = rng.standard_normal((200, 260, 512)).astype(np.float32)
    xd = gpu.asarray(x)
    a, b, c = rng.standard_normal(3), rng.standard_normal(3), rng.standard_normal(3)
    w = a[:, None, None] * b[None, :, None] * c[None, None, :]
    got = ndi.correlate(xd, w, mode="mirror").get()
    t = ndi.correlate1d(xd, a, axis=0, mode="mirror", dtype_mode="ndimage")
    t = ndi.correlate1d(t, b, axis=1, mode="mirror", dtype_mode="ndimage")
    ref = ndi.correlate1d(t, c, axis=2, mode="mirror", dtype_mode="ndimage").get()
    assert maxnorm_rel(got, ref) <= 2e-6


# ------------------------------------------------------------------ 2-D images through the fused kernels
@pytest.mark.parametrize("shape", [(300, 512), (64, 264), (33, 40), (1000, 1024)])
def test_2d_images_take_the_fused_kernels(gpu, ndi, shape):
    """A float32 image is filtered as a one-plane volume by the same fused /
    streaming kernels (no z taps); results match the oracle like the 3-D case."""
    rng = np.random.default_rng(50)
    x = rng.standard_normal(shape).astype(np.float32)
    xd = gpu.asarray(x)
    for mode in MODES:
        for size in (3, 5, (9, 3), (1, 7)):
            ref = orc.uniform_filter(x, size, mode=mode, cval=0.25)
            got = ndi.uniform_filter(xd, size, mode=mode, cval=0.25).get()
            assert maxnorm_rel(got, ref) <= 1e-6, (shape, "uniform", size, mode)
        for sigma in (0.8, 2.0, 3.0, (2.0, 0.7)):      # 7, 17 (x fused into the y pass), 25 taps, mixed
            if min(shape) < 30 and np.max(sigma) > 2.5:
                continue
            ref = orc.gaussian_filter(x, sigma, mode=mode, cval=0.25)
            got = ndi.gaussian_filter(xd, sigma, mode=mode, cval=0.25).get()
            assert maxnorm_rel(got, ref) <= 1e-6, (shape, "gaussian", sigma, mode)
    out = gpu.empty(shape, np.float32)
    assert ndi.gaussian_filter(xd, 1.0, output=out) is out
    assert maxnorm_rel(out.get(), orc.gaussian_filter(x, 1.0)) <= 1e-6


# ------------------------------------------------------------------ LDS-tiled binary morphology
@pytest.mark.parametrize("shape", [(20, 37, 64), (9, 50, 1040), (5, 16, 16), (70, 128), (33, 18, 2064), (12, 21, 520),
                                   (6, 9, 260)])
def test_tiled_binary_morphology_matches_generic_kernel_and_oracle(gpu, ndi, shape):
    """binary erosion / dilation on 1-byte volumes: the byte-parallel LDS-tiled
    kernel (binary3d.hip) gives exactly what the generic kernel and the oracle give."""
    import ctypes
    from cupyimg_amd import _lib
    lib = _lib.load()
    lib.mi_debug_set_binary_tiled.argtypes = [ctypes.c_int]
    rng = np.random.default_rng(60)
    nd = len(shape)
    x = rng.random(shape) > 0.35
    m = rng.random(shape) > 0.3
    xd, md = gpu.asarray(x), gpu.asarray(m)
    structs = [None, np.ones((3,) * nd, bool), rng.random((5, 3, 7)[:nd]) > 0.4, np.ones((2, 4, 3)[:nd], bool),
               rng.random((3, 9, 9)[3 - nd:]) > 0.5]
    for st in structs:
        for fn, ofn in [(ndi.binary_erosion, orc.binary_erosion), (ndi.binary_dilation, orc.binary_dilation)]:
            for kw in [dict(), dict(border_value=1), dict(iterations=3), dict(mask=True, iterations=2),
                       dict(origin=1 if st is None or min(st.shape) >= 3 else 0)]:
                kg = dict(kw)
                ko = dict(kw)
                if kw.get("mask"):
                    kg["mask"], ko["mask"] = md, m
                try:
                    lib.mi_debug_set_binary_tiled(1)
                    tiled = fn(xd, st, **kg).get()
                    lib.mi_debug_set_binary_tiled(0)
                    generic = fn(xd, st, **kg).get()
                finally:
                    lib.mi_debug_set_binary_tiled(1)
                ref = ofn(x, st, **ko)
                assert np.array_equal(generic, ref), (fn.__name__, kw)
                assert np.array_equal(tiled, ref), (fn.__name__, None if st is None else st.shape, kw)
    # uint8 volumes with values other than 0 / 1, iterate until stable
    u = (rng.random(shape) > 0.2).astype(np.uint8) * rng.integers(1, 255, size=shape, dtype=np.uint8)
    got = ndi.binary_erosion(gpu.asarray(u), iterations=-1).get()
    assert np.array_equal(got, orc.binary_erosion(u, iterations=-1))
    import scipy.ndimage as sndi
    got = ndi.binary_fill_holes(gpu.asarray(u)).get()
    assert np.array_equal(got, sndi.binary_fill_holes(u))


@pytest.mark.parametrize("case", range(6))
def test_tiled_footprint_minmax_matches_generic_kernel_and_oracle(gpu, ndi, case):
    """minimum / maximum filter and flat grey erosion / dilation with a footprint
    on float32: the LDS-tiled kernel compares like the generic one (NaNs included)."""
    import ctypes
    from cupyimg_amd import _lib
    lib = _lib.load()
    lib.mi_debug_set_minmax_tiled.argtypes = [ctypes.c_int]
    rng = np.random.default_rng(70 + case)
    shape, fshape, origin = [((20, 37, 64), (3, 3, 3), 0), ((9, 50, 264), (5, 3, 5), (1, -1, 0)),
                             ((17, 40, 256), (2, 4, 6), (0, 1, -1)), ((8, 20, 72), (7, 7, 7), 0),
                             ((40, 300), (5, 5), 0), ((12, 30, 128), (3, 1, 9), (0, 0, 2))][case]
    x = rng.standard_normal(shape).astype(np.float32)
    x[tuple(s // 3 for s in shape)] = np.nan
    x[tuple(s // 2 for s in shape)] = np.inf
    fp = rng.random(fshape) > 0.35
    fp[tuple(s // 2 for s in fshape)] = True
    for fp_ in (fp, np.ones(fshape, bool)):
        if fp_.all():
            # an all-ones footprint is filtered separably (one 1-D pass per axis, like SciPy); the order in
            # which a NaN meets the comparisons then depends on the pass order, so no NaN in this leg
            x = np.where(np.isnan(x), np.float32(0.5), x)
        xd = gpu.asarray(x)
        for fn, ofn in [(ndi.minimum_filter, orc.minimum_filter), (ndi.maximum_filter, orc.maximum_filter),
                        (ndi.grey_erosion, orc.grey_erosion), (ndi.grey_dilation, orc.grey_dilation)]:
            for mode in MODES:
                try:
                    lib.mi_debug_set_minmax_tiled(1)
                    tiled = fn(xd, footprint=fp_, mode=mode, cval=0.3, origin=origin).get()
                    lib.mi_debug_set_minmax_tiled(0)
                    generic = fn(xd, footprint=fp_, mode=mode, cval=0.3, origin=origin).get()
                finally:
                    lib.mi_debug_set_minmax_tiled(1)
                ref = ofn(x, footprint=fp_, mode=mode, cval=0.3, origin=origin)
                assert np.array_equal(generic, ref, equal_nan=True), (fn.__name__, mode)
                assert np.array_equal(tiled, ref, equal_nan=True), (fn.__name__, mode, fp_.all())


@pytest.mark.parametrize("shape", [(20, 37, 64), (33, 21, 264), (9, 40, 512), (70, 300), (12, 5, 8)])
def test_streaming_minmax_f32_is_exact(gpu, ndi, shape):
    """Separable flat min / max (minimum/maximum_filter, grey erosion/dilation
    with `size`) on float32: streaming passes, results equal to the oracle."""
    rng = np.random.default_rng(80)
    x = rng.standard_normal(shape).astype(np.float32)
    xd = gpu.asarray(x)
    nd = len(shape)
    cases = [(3, 0), (5, 0), (7, 0), (9, 0), ((3, 5, 7)[:nd], 0), ((1, 9, 3)[3 - nd:], 0), ((5, 5, 3)[:nd], ((1, -2, 0)[:nd]))]
    for size, origin in cases:
        if np.max(size) > min(shape) * 2:
            continue
        for fn, ofn in [(ndi.minimum_filter, orc.minimum_filter), (ndi.maximum_filter, orc.maximum_filter),
                        (ndi.grey_erosion, orc.grey_erosion), (ndi.grey_dilation, orc.grey_dilation)]:
            for mode in MODES:
                got = fn(xd, size=size, mode=mode, cval=0.3, origin=origin).get()
                ref = ofn(x, size=size, mode=mode, cval=0.3, origin=origin)
                assert np.array_equal(got, ref), (fn.__name__, size, origin, mode)
    # per-axis modes
    modes = ["nearest", "wrap", "mirror"][:nd]
    got = ndi.maximum_filter(xd, size=5, mode=modes).get()
    assert np.array_equal(got, orc.maximum_filter(x, size=5, mode=modes))


@pytest.mark.parametrize("shape", [(20, 37, 64), (9, 50, 1040), (9, 40, 2064), (40, 16, 128), (5, 3, 1024)])
def test_fused_uint8_minmax_matches_two_launch_path_and_oracle(gpu, ndi, shape):
    """grey erosion / dilation, min / max filter with cubic size 3 / 5 / 7 on
    uint8: the single-launch kernel equals the two-launch path and the oracle."""
    import ctypes
    from cupyimg_amd import _lib
    lib = _lib.load()
    lib.mi_debug_set_u8_fused.argtypes = [ctypes.c_int]
    rng = np.random.default_rng(90)
    x = rng.integers(0, 256, size=shape, dtype=np.uint8)
    xd = gpu.asarray(x)
    for size in (3, 5, 7):
        for fn, ofn in [(ndi.grey_erosion, orc.grey_erosion), (ndi.maximum_filter, orc.maximum_filter)]:
            for mode in MODES:
                try:
                    lib.mi_debug_set_u8_fused(1)
                    fused = fn(xd, size=size, mode=mode, cval=37).get()
                    lib.mi_debug_set_u8_fused(0)
                    two = fn(xd, size=size, mode=mode, cval=37).get()
                finally:
                    lib.mi_debug_set_u8_fused(1)
                ref = ofn(x, size=size, mode=mode, cval=37)
                assert np.array_equal(two, ref), (fn.__name__, size, mode)
                assert np.array_equal(fused, ref), (fn.__name__, size, mode)
    modes = ["nearest", "constant", "mirror"]
    got = ndi.grey_erosion(xd, size=7, mode=modes, cval=200).get()
    assert np.array_equal(got, orc.grey_erosion(x, size=7, mode=modes, cval=200))


@pytest.mark.parametrize("dtype", ["uint8", "uint16", "int16"])
def test_tiled_stencil_and_footprint_minmax_integer_volumes(gpu, ndi, dtype):
    """8 / 16-bit integer volumes on the LDS-tiled kernels (values converted to
    float at staging, cast back on store): identical to the generic kernels and
    the oracle, including the truncating / wrapping output cast of correlate."""
    import ctypes
    from cupyimg_amd import _lib
    lib = _lib.load()
    lib.mi_debug_set_stencil.argtypes = [ctypes.c_int]
    lib.mi_debug_set_minmax_tiled.argtypes = [ctypes.c_int]
    rng = np.random.default_rng(100)
    info = np.iinfo(dtype)
    for shape, wshape, origin in [((20, 37, 64), (3, 3, 3), 0), ((9, 50, 264), (5, 3, 5), (1, -1, 0)), ((40, 300), (5, 5), 0),
                                  ((17, 40, 256), (2, 4, 6), (0, 1, -1))]:
        x = rng.integers(info.min, info.max + 1, size=shape).astype(dtype)
        xd = gpu.asarray(x)
        w = rng.standard_normal(wshape) * 0.2
        fp = rng.random(wshape) > 0.35
        fp[tuple(s // 2 for s in wshape)] = True
        for mode in MODES:
            try:
                lib.mi_debug_set_stencil(1); lib.mi_debug_set_minmax_tiled(1)
                t_corr = ndi.correlate(xd, w, mode=mode, cval=7, origin=origin).get()
                t_conv = ndi.convolve(xd, w, mode=mode, cval=7, origin=origin).get()
                t_min = ndi.minimum_filter(xd, footprint=fp, mode=mode, cval=7, origin=origin).get()
                t_max = ndi.grey_dilation(xd, footprint=fp, mode=mode, cval=7, origin=origin).get()
                lib.mi_debug_set_stencil(0); lib.mi_debug_set_minmax_tiled(0)
                g_corr = ndi.correlate(xd, w, mode=mode, cval=7, origin=origin).get()
                g_min = ndi.minimum_filter(xd, footprint=fp, mode=mode, cval=7, origin=origin).get()
            finally:
                lib.mi_debug_set_stencil(1); lib.mi_debug_set_minmax_tiled(1)
            assert t_corr.dtype == x.dtype
            assert np.array_equal(t_corr, g_corr), (shape, mode)
            assert np.array_equal(t_min, g_min), (shape, mode)
            assert np.array_equal(t_corr, orc.correlate(x, w, mode=mode, cval=7, origin=origin)), (shape, mode)
            assert np.array_equal(t_conv, orc.convolve(x, w, mode=mode, cval=7, origin=origin)), (shape, mode)
            assert np.array_equal(t_min, orc.minimum_filter(x, footprint=fp, mode=mode, cval=7, origin=origin)), (shape, mode)
            assert np.array_equal(t_max, orc.grey_dilation(x, footprint=fp, mode=mode, cval=7, origin=origin)), (shape, mode)


def test_affine_and_map_2d_images_take_the_fast_kernels(gpu, ndi):
    """float32 images are one-plane volumes for the order 0 / 1 gather kernels
    (z coordinate exactly 0); tolerance as for 3-D volumes: 2e-6 * max|ref|."""
    rng = np.random.default_rng(16)
    x = rng.standard_normal((150, 203)).astype(np.float32)
    ang = np.deg2rad(11.0)
    M = np.array([[1.03 * np.cos(ang), -np.sin(ang)], [np.sin(ang), 0.97 * np.cos(ang)]])
    off = np.array([3.5, -7.25])
    xd = gpu.asarray(x)
    for mode in ["constant", "grid-constant", "nearest", "mirror", "reflect", "wrap", "grid-wrap"]:
        for order in [0, 1]:
            ref = orc.affine_transform(x, M, off, order=order, mode=mode, cval=0.25)
            got = ndi.affine_transform(xd, M, off, order=order, mode=mode, cval=0.25).get()
            assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (mode, order)
            ref = orc.affine_transform(x, M, off, output_shape=(64, 300), order=order, mode=mode, cval=0.25)
            got = ndi.affine_transform(xd, M, off, output_shape=(64, 300), order=order, mode=mode, cval=0.25).get()
            assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (mode, order)
    idx = np.indices(x.shape).reshape(2, -1).astype(np.float64)
    for cdt in (np.float32, np.float64):
        coords = (M @ idx + off[:, None]).reshape((2,) + x.shape).astype(cdt)
        for mode in ["constant", "reflect", "wrap"]:
            for order in [0, 1]:
                ref = orc.map_coordinates(x, coords, order=order, mode=mode, cval=-1.0)
                got = ndi.map_coordinates(xd, gpu.asarray(coords), order=order, mode=mode, cval=-1.0).get()
                assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (cdt, mode, order)


# ------------------------------------------------------------------ B-spline orders 2-5
SPLINE_MODES = ["constant", "grid-constant", "nearest", "mirror", "reflect", "wrap", "grid-wrap"]


def _close(got, ref, tol=1e-9):
    return np.abs(np.asarray(got, dtype=np.float64) - ref).max() <= tol * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("shape", [(23,), (17, 19), (9, 10, 11), (1, 12), (6, 1, 7)])
def test_spline_filter_matches_oracle(gpu, ndi, shape):
    rng = np.random.default_rng(110)
    x = rng.standard_normal(shape)
    xd = gpu.asarray(x)
    for order in (2, 3, 4, 5):
        for mode in SPLINE_MODES + ["grid-mirror"]:
            assert _close(ndi.spline_filter(xd, order, mode=mode).get(), orc.spline_filter(x, order, mode=mode)), (order, mode)
            for ax in range(len(shape)):
                got = ndi.spline_filter1d(xd, order, axis=ax, mode=mode).get()
                assert _close(got, orc.spline_filter1d(x, order, ax, mode=mode)), (order, mode, ax)
    assert ndi.spline_filter(gpu.asarray(x.astype(np.float32)), 3, output=np.float32).dtype == np.float32
    assert np.array_equal(ndi.spline_filter1d(xd, 1).get(), x)
    with pytest.raises(RuntimeError):
        ndi.spline_filter(xd, 6)


@pytest.mark.parametrize("shape", [(40,), (21, 26), (9, 12, 14)])
@pytest.mark.parametrize("order", [2, 3, 4, 5])
def test_spline_interpolation_matches_oracle(gpu, ndi, shape, order):
    """map_coordinates / affine_transform / shift / zoom with spline orders 2-5:
    device coefficients (pad + prefilter) and tap loop against the oracle, which
    restates SciPy 1.15.3 (agreement 1e-14 there)."""
    rng = np.random.default_rng(111)
    nd = len(shape)
    x = rng.standard_normal(shape).astype(np.float32)
    xd = gpu.asarray(x)
    coords = rng.uniform(-20, max(shape) + 20, size=(nd, 300))
    coords[:, 0] = 0.0
    coords[:, 1] = [n - 1 for n in shape]
    cdev = gpu.asarray(coords)
    M = np.eye(nd) * 0.93 + rng.standard_normal((nd, nd)) * 0.08
    off = rng.standard_normal(nd) * 2
    for mode in SPLINE_MODES:
        for pf in (True, False):
            ref = orc.map_coordinates(x.astype(np.float64), coords, order=order, mode=mode, cval=0.7, prefilter=pf)
            got = ndi.map_coordinates(xd, cdev, order=order, mode=mode, cval=0.7, prefilter=pf, output=np.float64).get()
            assert _close(got, ref), ("map", mode, pf)
        ref = orc.affine_transform(x.astype(np.float64), M, off, order=order, mode=mode, cval=0.7)
        got = ndi.affine_transform(xd, M, off, order=order, mode=mode, cval=0.7, output=np.float64).get()
        assert _close(got, ref), ("affine", mode)
        # float32 in / out like a user would call it: same values rounded to float32
        got32 = ndi.affine_transform(xd, M, off, order=order, mode=mode, cval=0.7).get()
        assert got32.dtype == np.float32 and _close(got32, ref, 1e-6), ("affine f32", mode)
        ref = orc.shift(x.astype(np.float64), 1.3, order=order, mode=mode, cval=0.7)
        assert _close(ndi.shift(xd, 1.3, order=order, mode=mode, cval=0.7, output=np.float64).get(), ref), ("shift", mode)
        for gm in (False, True):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref = orc.zoom(x.astype(np.float64), 1.7, order=order, mode=mode, cval=0.7, grid_mode=gm)
                got = ndi.zoom(xd, 1.7, order=order, mode=mode, cval=0.7, grid_mode=gm, output=np.float64).get()
            assert got.shape == ref.shape and _close(got, ref), ("zoom", mode, gm)


def test_spline_defaults_integers_and_rotate(gpu, ndi):
    """default order 3 (what scipy.ndimage callers get), integer images (rounded
    output), rotate against scipy.ndimage itself"""
    import scipy.ndimage as sndi
    rng = np.random.default_rng(112)
    u = rng.integers(0, 256, size=(31, 37)).astype(np.uint8)
    ud = gpu.asarray(u)
    assert np.array_equal(ndi.zoom(ud, 2.0).get(), sndi.zoom(u, 2.0))
    assert np.array_equal(ndi.shift(ud, (1.5, -2.25)).get(), sndi.shift(u, (1.5, -2.25)))
    x = rng.standard_normal((24, 30))
    xd = gpu.asarray(x)
    for angle, reshape in [(30.0, True), (90.0, True), (-17.5, False), (180.0, False)]:
        for order in (0, 1, 3):
            ref = sndi.rotate(x, angle, reshape=reshape, order=order, mode="nearest")
            got = ndi.rotate(xd, angle, reshape=reshape, order=order, mode="nearest").get()
            assert got.shape == ref.shape and _close(got, ref, 1e-9), (angle, reshape, order)
    v = rng.standard_normal((7, 12, 10))
    ref = sndi.rotate(v, 25.0, axes=(2, 0), order=3, mode="mirror")
    got = ndi.rotate(gpu.asarray(v), 25.0, axes=(2, 0), order=3, mode="mirror").get()
    assert got.shape == ref.shape and _close(got, ref, 1e-9)
    c = rng.uniform(0, 20, size=(2, 50))
    assert _close(ndi.map_coordinates(xd, gpu.asarray(c)).get(), sndi.map_coordinates(x, c))     # defaults: order 3, constant


# ------------------------------------------------------------------ composite filters (derivatives, top-hats)
@pytest.mark.parametrize("dtype", ["float32", "float64", "uint8", "int16"])
def test_derivative_and_morphological_composites_match_scipy(gpu, ndi, dtype):
    """prewitt / sobel / laplace / gaussian_laplace / gaussian_gradient_magnitude and
    the grey-morphology composites are chains of the path's filters plus
    element-wise device arithmetic (mi_elementwise); compared with scipy.ndimage."""
    import scipy.ndimage as sndi
    rng = np.random.default_rng(120)
    for shape in [(20, 33), (9, 14, 16)]:
        if np.dtype(dtype).kind == "f":
            x = rng.standard_normal(shape).astype(dtype)
        else:
            x = rng.integers(0, 100, size=shape).astype(dtype)
        xd = gpu.asarray(x)
        tol = 0 if np.dtype(dtype).kind in "iu" else (2e-6 if dtype == "float32" else 1e-12)

        def same(got, ref, what):
            got = got.get()
            assert got.dtype == ref.dtype and got.shape == ref.shape, what
            if tol == 0:
                assert np.array_equal(got, ref), what
            else:
                assert np.abs(got.astype(np.float64) - ref).max() <= tol * max(1.0, np.abs(ref).max()), what
        for mode in ["reflect", "constant", "nearest", "mirror", "wrap"]:
            for ax in range(len(shape)):
                same(ndi.prewitt(xd, axis=ax, mode=mode, cval=2.0), sndi.prewitt(x, axis=ax, mode=mode, cval=2.0), ("prewitt", mode, ax))
                same(ndi.sobel(xd, axis=ax, mode=mode, cval=2.0), sndi.sobel(x, axis=ax, mode=mode, cval=2.0), ("sobel", mode, ax))
            same(ndi.laplace(xd, mode=mode, cval=2.0), sndi.laplace(x, mode=mode, cval=2.0), ("laplace", mode))
            if np.dtype(dtype).kind == "f":
                same(ndi.gaussian_laplace(xd, 1.0, mode=mode), sndi.gaussian_laplace(x, 1.0, mode=mode), ("glaplace", mode))
                same(ndi.gaussian_gradient_magnitude(xd, 1.2, mode=mode), sndi.gaussian_gradient_magnitude(x, 1.2, mode=mode),
                     ("ggm", mode))
            for fn, sfn in [(ndi.morphological_gradient, sndi.morphological_gradient), (ndi.morphological_laplace, sndi.morphological_laplace),
                            (ndi.white_tophat, sndi.white_tophat), (ndi.black_tophat, sndi.black_tophat),
                            (ndi.grey_opening, sndi.grey_opening), (ndi.grey_closing, sndi.grey_closing)]:
                same(fn(xd, size=3, mode=mode, cval=2.0), sfn(x, size=3, mode=mode, cval=2.0), (fn.__name__, mode))
    b = rng.random((20, 33)) > 0.5
    fp = np.ones((3, 3), bool)
    assert np.array_equal(ndi.white_tophat(gpu.asarray(b), footprint=fp).get(), sndi.white_tophat(b, footprint=fp))
    assert np.array_equal(ndi.black_tophat(gpu.asarray(b), footprint=fp).get(), sndi.black_tophat(b, footprint=fp))
    assert np.array_equal(ndi.binary_hit_or_miss(gpu.asarray(b)).get(), sndi.binary_hit_or_miss(b))
    assert np.array_equal(ndi.binary_fill_holes(gpu.asarray(b)).get(), sndi.binary_fill_holes(b))


def test_derivative_filters_take_dtype_mode(gpu, ndi):
    """prewitt / sobel / laplace accept the reference's keyword-only `dtype_mode`
    (cupyimg/scipy/ndimage/filters.py:828-838, 889-899, 1041-1043) and hand it to the
    1-D passes: "ndimage" = SciPy's double accumulation (exact on integers),
    "float" = float32 accumulation for <= 32-bit inputs; anything else is a ValueError."""
    import scipy.ndimage as sndi
    rng = np.random.default_rng(121)
    xi = rng.integers(0, 100, size=(20, 33)).astype(np.int16)
    xf = rng.standard_normal((9, 14, 16)).astype(np.float32)
    for x, tol in [(xi, 0), (xf, 2e-6)]:
        xd = gpu.asarray(x)
        for dm in ("ndimage", "float"):
            for got, ref in [(ndi.prewitt(xd, axis=0, dtype_mode=dm), sndi.prewitt(x, axis=0)),
                             (ndi.sobel(xd, axis=1, mode="mirror", dtype_mode=dm), sndi.sobel(x, axis=1, mode="mirror")),
                             (ndi.laplace(xd, mode="nearest", dtype_mode=dm), sndi.laplace(x, mode="nearest"))]:
                got = got.get()
                assert got.dtype == ref.dtype
                if tol == 0:
                    assert np.array_equal(got, ref), dm       # small integers: float32 sums are exact too
                else:
                    assert np.abs(got.astype(np.float64) - ref).max() <= tol * max(1.0, np.abs(ref).max()), dm
        for fn in (ndi.prewitt, ndi.sobel, ndi.laplace):
            with pytest.raises(ValueError):
                fn(xd, dtype_mode="bogus")
            with pytest.raises(TypeError):
                fn(xd, None, None, "reflect", 0.0, "float") if fn is not ndi.laplace else fn(xd, None, "reflect", 0.0, "float")


@pytest.mark.parametrize("dtype", ["float32", "uint8", "int16", "float64"])
def test_rank_median_percentile_filters_match_scipy(gpu, ndi, dtype):
    import scipy.ndimage as sndi
    rng = np.random.default_rng(130)
    for shape, size in [((30, 41), 3), ((30, 41), (5, 3)), ((9, 14, 16), 3), ((64,), 5)]:
        x = rng.standard_normal(shape).astype(dtype) if np.dtype(dtype).kind == "f" else rng.integers(0, 200, size=shape).astype(dtype)
        xd = gpu.asarray(x)
        for mode in ["reflect", "constant", "nearest", "mirror", "wrap"]:
            assert np.array_equal(ndi.median_filter(xd, size=size, mode=mode, cval=3).get(), sndi.median_filter(x, size=size, mode=mode, cval=3))
            assert np.array_equal(ndi.rank_filter(xd, 2, size=size, mode=mode, cval=3).get(), sndi.rank_filter(x, 2, size=size, mode=mode, cval=3))
            assert np.array_equal(ndi.rank_filter(xd, -2, size=size, mode=mode, cval=3).get(), sndi.rank_filter(x, -2, size=size, mode=mode, cval=3))
            for pct in (0, 20, 50, 75.5, 100, -30):
                assert np.array_equal(ndi.percentile_filter(xd, pct, size=size, mode=mode, cval=3).get(),
                                      sndi.percentile_filter(x, pct, size=size, mode=mode, cval=3)), (pct, mode)
    fp = rng.random((3, 5)) > 0.4
    fp[1, 2] = True
    x = rng.standard_normal((25, 28)).astype(np.float32)
    got = ndi.median_filter(gpu.asarray(x), footprint=fp, origin=(0, 1), mode="mirror").get()
    assert np.array_equal(got, sndi.median_filter(x, footprint=fp, origin=(0, 1), mode="mirror"))
    with pytest.raises(RuntimeError):
        ndi.rank_filter(gpu.asarray(x), 99, size=3)


def test_float32_cubic_route_matches_scipy(gpu, ndi):
    """float32 image -> float32 result at order 3 stores float32 coefficients and gathers in
    float32 (allow_float32, the reference's interpolation.py:330-335); SciPy works in double,
    so the comparison carries a tolerance: 2e-5 of the data range (the survey proposes 1e-5
    abs/rel for identical arithmetic; the reference itself claims 1e-4)."""
    import scipy.ndimage as sndi
    rng = np.random.default_rng(170)
    modes = ["constant", "nearest", "mirror", "reflect", "wrap", "grid-wrap", "grid-constant", "grid-mirror"]
    for shape in [(33, 38), (14, 17, 19), (41,)]:
        x = rng.standard_normal(shape).astype(np.float32)
        xd = gpu.asarray(x)
        nd = len(shape)
        coords = (rng.random((nd, 2000)) * (np.array(shape)[:, None] + 6) - 3).astype(np.float32)
        # exact grid points and edges too
        coords[:, :50] = np.round(coords[:, :50])
        A = np.eye(nd) + 0.1 * rng.standard_normal((nd, nd))
        off = rng.standard_normal(nd)
        tol = 2e-5 * float(np.abs(x).max())
        for mode in modes:
            for prefilter in (True, False):
                want = sndi.map_coordinates(x.astype(np.float64), coords.astype(np.float64), order=3, mode=mode, cval=0.5,
                                            prefilter=prefilter)
                got = ndi.map_coordinates(xd, gpu.asarray(coords), order=3, mode=mode, cval=0.5, prefilter=prefilter)
                assert got.dtype == np.float32
                assert np.abs(got.get() - want).max() <= tol * 4, (shape, mode, prefilter)
            want = sndi.affine_transform(x.astype(np.float64), A, offset=off, order=3, mode=mode, cval=-1.0)
            got = ndi.affine_transform(xd, A, offset=off, order=3, mode=mode, cval=-1.0)
            assert got.dtype == np.float32
            assert np.abs(got.get() - want).max() <= tol * 4, (shape, mode)
            # the double route stays available and is much closer
            exact = ndi.affine_transform(xd, A, offset=off, order=3, mode=mode, cval=-1.0, allow_float32=False)
            assert np.abs(exact.get() - want).max() <= 2e-6 * float(np.abs(x).max()), (shape, mode)
        for mode in ("constant", "nearest", "reflect"):
            want = sndi.shift(x.astype(np.float64), 1.7, order=3, mode=mode)
            assert np.abs(ndi.shift(xd, 1.7, order=3, mode=mode).get() - want).max() <= tol * 4
            want = sndi.zoom(x.astype(np.float64), 1.3, order=3, mode=mode)
            assert np.abs(ndi.zoom(xd, 1.3, order=3, mode=mode).get() - want).max() <= tol * 4
    img = rng.standard_normal((64, 80)).astype(np.float32)
    want = sndi.rotate(img.astype(np.float64), 17.0, order=3, reshape=True)
    got = ndi.rotate(gpu.asarray(img), 17.0, order=3, reshape=True)
    assert got.shape == want.shape and np.abs(got.get() - want).max() <= 8e-5 * float(np.abs(img).max())
    # in-place request and a float64 output keep working
    out64 = gpu.empty(img.shape, np.float64)
    ndi.affine_transform(gpu.asarray(img), np.eye(2) * 0.9, order=3, output=out64)
    np.testing.assert_allclose(out64.get(), sndi.affine_transform(img.astype(np.float64), np.eye(2) * 0.9, order=3),
                               rtol=1e-9, atol=1e-9)


def test_spline_prefilter_contiguous_lines_kernel(gpu, ndi):
    """Lines along the last axis (>= 128 samples) go through the kernel that holds whole lines in LDS (r4b: twelve per
    wave, the one-thread-per-line kernel's own code on the copy -- bit-identical to it) or, when they do not fit / with
    the debug knob at 0, the LDS-tiled kernel: same results as the one-thread-per-line kernel and as SciPy, every order and
    boundary rule, ragged sizes, float64 and float32 coefficients, other line counts per wave."""
    import ctypes
    import scipy.ndimage as sndi
    from cupyimg_amd import _lib
    hook = _lib.load().mi_debug_set_spline_rows
    hook.argtypes = [ctypes.c_int]
    lds_hook = _lib.load().mi_debug_set_spline_rows_lds
    rng = np.random.default_rng(171)
    hook(2)         # the kernels are normally chosen for >= 16384 lines; force them for the small cases
    try:
        for lds in (0, 1, 5, 64):
            lds_hook(lds)
            for shape in [(70, 130), (3, 5, 257), (129,), (64, 128)]:
                x = rng.standard_normal(shape)
                for order in (2, 3, 4, 5):
                    for mode in ("mirror", "reflect", "grid-wrap", "nearest", "constant", "wrap"):
                        want = sndi.spline_filter1d(x, order, axis=-1, mode=mode)
                        got = ndi.spline_filter1d(gpu.asarray(x), order, axis=-1, mode=mode).get()
                        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12, err_msg=str((shape, order, mode)))
                        hook(0)
                        old = ndi.spline_filter1d(gpu.asarray(x), order, axis=-1, mode=mode).get()
                        hook(2)
                        if lds:
                            assert np.array_equal(got, old), (lds, shape, order, mode)
                        else:
                            np.testing.assert_allclose(got, old, rtol=1e-14, atol=1e-14, err_msg=str((shape, order, mode)))
                xf = x.astype(np.float32)
                got = ndi.spline_filter(gpu.asarray(xf), 3, output=np.float32, allow_float32=True)
                assert got.dtype == np.float32
                np.testing.assert_allclose(got.get(), sndi.spline_filter(xf.astype(np.float64), 3), rtol=0,
                                           atol=2e-6 * np.abs(xf).max())
    finally:
        hook(1)
        lds_hook(1)
    # the default choice on a volume with many lines
    v = rng.standard_normal((130, 130, 140)).astype(np.float32)
    got = ndi.spline_filter(gpu.asarray(v), 3, output=np.float64)
    np.testing.assert_allclose(got.get(), sndi.spline_filter(v.astype(np.float64), 3), rtol=1e-11, atol=1e-11)


def test_float32_cubic_zoom_shift_strip_kernel(gpu, ndi):
    """Diagonal transforms on the float32 cubic route run as separable 1-D resampling passes (default) or
    blend rows into an LDS strip (one launch); same results as the gather kernel (float rounding apart)
    and SciPy within the route's tolerance."""
    import ctypes
    import scipy.ndimage as sndi
    from cupyimg_amd import _lib
    hook = _lib.load().mi_debug_set_cubic_diag
    hook.argtypes = [ctypes.c_int]
    sep_hook = _lib.load().mi_debug_set_cubic_separable
    sep_hook.argtypes = [ctypes.c_int]
    rng = np.random.default_rng(172)
    modes = ["constant", "nearest", "mirror", "reflect", "wrap", "grid-wrap", "grid-constant"]
    for shape in [(70, 200), (9, 33, 150), (300,)]:
        x = rng.standard_normal(shape).astype(np.float32)
        xd = gpu.asarray(x)
        tol = 8e-5 * float(np.abs(x).max())
        for mode in modes:
            for zf in (0.37, 0.8, 1.0, 1.25, 2.6):
                want = sndi.zoom(x.astype(np.float64), zf, order=3, mode=mode)
                got = ndi.zoom(xd, zf, order=3, mode=mode)
                assert got.shape == want.shape
                assert np.abs(got.get() - want).max() <= tol, (shape, mode, zf)
                hook(0)
                try:
                    ref = ndi.zoom(xd, zf, order=3, mode=mode).get()
                finally:
                    hook(1)
                assert np.abs(got.get() - ref).max() <= 4e-6 * float(np.abs(x).max()), (shape, mode, zf)
                sep_hook(0)         # the one-launch LDS strip kernel instead of the separable passes
                try:
                    strip = ndi.zoom(xd, zf, order=3, mode=mode).get()
                finally:
                    sep_hook(1)
                assert np.abs(strip - ref).max() <= 2e-6 * float(np.abs(x).max()), (shape, mode, zf)
            for sh in (1.7, -3.25, [0.5, -2.0, 4.75][:len(shape)]):
                want = sndi.shift(x.astype(np.float64), sh, order=3, mode=mode, cval=2.0)
                got = ndi.shift(xd, sh, order=3, mode=mode, cval=2.0)
                assert np.abs(got.get() - want).max() <= tol, (shape, mode, sh)
        # diagonal affine with a flip (negative x step)
        d = np.array([1.1, -0.9, 0.7][:len(shape)])
        off = np.where(d < 0, np.array(shape) - 1.0, 0.0)
        want = sndi.affine_transform(x.astype(np.float64), d, offset=off, order=3, mode="mirror")
        got = ndi.affine_transform(xd, d, offset=off, order=3, mode="mirror")
        assert np.abs(got.get() - want).max() <= tol


def test_rank_filter_sorting_network_sizes(gpu, ndi):
    """Footprints of 9 ... 64 samples take the register sorting network (three padded sizes); beyond that and
    for wide dtypes the selection kernel.  Both against SciPy."""
    import scipy.ndimage as sndi
    rng = np.random.default_rng(131)
    img = rng.integers(0, 255, size=(45, 52)).astype(np.uint8)
    vol = rng.standard_normal((12, 20, 22)).astype(np.float32)
    for x in (img, vol, img.astype(np.uint16) * 200, vol.astype(np.float64), img.astype(np.int32) - 100):
        xd = gpu.asarray(x)
        sizes = [3, 4, 5, 7, 8, 9] if x.ndim == 2 else [2, 3, 4]
        for size in sizes:
            n = size ** x.ndim
            for rank in sorted({0, 1, n // 3, n // 2, n - 2, n - 1}):
                want = sndi.rank_filter(x, rank, size=size, mode="reflect")
                assert np.array_equal(ndi.rank_filter(xd, rank, size=size, mode="reflect").get(), want), (x.dtype, size, rank)
        assert np.array_equal(ndi.median_filter(xd, size=3, mode="constant", cval=7).get(),
                              sndi.median_filter(x, size=3, mode="constant", cval=7))


def test_regressions_found_by_differential_fuzzing(gpu, ndi):
    """Cases scripts/fuzz_vs_scipy.py caught: float32 separable min/max on rows of 256 k + 4 samples (a one-lane
    last tile in the streaming x pass), and order 0/1 interpolation in wrap mode across an axis of length one."""
    import scipy.ndimage as sndi
    rng = np.random.default_rng(180)
    for nx in (260, 264, 268, 516, 520):
        x = rng.standard_normal((7, 19, nx)).astype(np.float32)
        xd = gpu.asarray(x)
        for size, origin in [((1, 3, 7), 0), ((7, 1, 5), (2, 0, 0)), ((3, 5, 7), (-1, 1, 0)), ((7, 7, 7), 0), ((1, 1, 3), 0)]:
            for mode in ("mirror", "reflect", "nearest", "constant", "wrap"):
                for f in ("minimum_filter", "maximum_filter"):
                    want = getattr(sndi, f)(x, size=size, origin=origin, mode=mode, cval=-2.0)
                    got = getattr(ndi, f)(xd, size=size, origin=origin, mode=mode, cval=-2.0).get()
                    assert np.array_equal(got, want), (nx, size, origin, mode, f)
        assert np.array_equal(ndi.grey_dilation(xd, size=(7, 7, 7), mode="mirror").get(), sndi.grey_dilation(x, size=(7, 7, 7), mode="mirror"))
    for shape, dtype in [((1, 34, 88), np.float32), ((7, 1, 8), np.float64), ((1, 2, 80), np.float32), ((1, 7, 256), np.int32),
                         ((1, 1024), np.float32)]:
        x = (rng.standard_normal(shape) * 50).astype(dtype)
        nd = len(shape)
        coords = rng.random((nd, 300)) * (np.array(shape)[:, None] + 4) - 2
        for mode in ("wrap", "grid-wrap", "mirror", "reflect", "nearest"):
            for order in (0, 1):
                cc = np.floor(coords * 4) / 4 + 0.1 if order == 0 else coords
                want = sndi.map_coordinates(x, cc, order=order, mode=mode)
                got = ndi.map_coordinates(gpu.asarray(x), gpu.asarray(cc), order=order, mode=mode).get()
                if np.dtype(dtype).kind == "f":
                    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4 * np.abs(x).max(), err_msg=str((shape, mode, order)))
                else:
                    assert np.abs(got.astype(np.int64) - want).max() <= 1, (shape, mode, order)
            want = sndi.shift(x, [0.3] * nd, order=1, mode=mode)
            got = ndi.shift(gpu.asarray(x), [0.3] * nd, order=1, mode=mode).get()
            assert np.abs(got.astype(np.float64) - want).max() <= max(1.0 if np.dtype(dtype).kind == "i" else 0.0, 1e-4 * np.abs(x).max())


def test_streaming_passes_on_many_workgroups(gpu, ndi):
    """Volumes whose streaming passes put several waves on a SIMD (the round-1 mid-size fuzz found wrong samples in
    lanes 12-15 of each 16-lane group: a VALU write overtook the data read of the preceding buffer_store_dwordx4 with a
    register soffset -- see buffer_store_b128_soff() in csrc/sep_common.hpp; passes run as ONE launch again)."""
    import scipy.ndimage as sndi
    rng = np.random.default_rng(190)
    for shape in [(256, 256, 256), (200, 300, 256), (150, 600, 64)]:
        v = rng.standard_normal(shape).astype(np.float32)
        vd = gpu.asarray(v)
        for size in [(3, 1, 3), (1, 3, 3), (3, 3, 3), (5, 3, 7)]:
            assert np.array_equal(ndi.minimum_filter(vd, size=size, mode="mirror").get(), sndi.minimum_filter(v, size=size, mode="mirror")), (shape, size)
        for sigma in (2.0, [1.0, 1.7, 1.0]):
            want = sndi.gaussian_filter(v, sigma)
            assert np.abs(ndi.gaussian_filter(vd, sigma).get() - want).max() <= 2e-6 * np.abs(want).max(), (shape, sigma)
        want = sndi.uniform_filter(v, size=(3, 5, 7))
        assert np.abs(ndi.uniform_filter(vd, size=(3, 5, 7)).get() - want).max() <= 2e-6 * np.abs(want).max()


def test_fused_long_kernel_against_scipy(gpu, ndi):
    """sep3d_long.hip (cubic 9..17 taps in ONE launch: LDS-DMA staging, y pass out of LDS, x in registers, z as a
    register scatter): every boundary mode (constant = zero fill + coverage correction), partial tiles in x and y,
    several z chunks, origins on y / z; the streaming passes (hook) must agree with it."""
    import ctypes
    import scipy.ndimage as sndi
    from cupyimg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(77)
    for shape in [(40, 48, 256), (33, 21, 260), (70, 100, 512), (20, 16, 64), (9, 9, 16), (5, 70, 300), (130, 40, 252)]:
        v = rng.standard_normal(shape).astype(np.float32)
        vd = gpu.asarray(v)
        for mode in ["reflect", "mirror", "nearest", "wrap"]:
            for size in (11, 17):
                want = sndi.uniform_filter(v.astype(np.float64), size, mode=mode)
                got = ndi.uniform_filter(vd, size, mode=mode).get()
                assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max(), (shape, mode, size)
            want = sndi.gaussian_filter(v.astype(np.float64), 1.6, mode=mode)      # 13 taps
            got = ndi.gaussian_filter(vd, 1.6, mode=mode).get()
            assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max(), (shape, mode)
        for mode, cval in [("constant", 1.25), (["constant", "mirror", "wrap"], -3.0)]:
            want = sndi.gaussian_filter(v.astype(np.float64), 2.0, mode=mode, cval=cval)
            got = ndi.gaussian_filter(vd, 2.0, mode=mode, cval=cval).get()
            assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max(), (shape, mode)
        want = sndi.uniform_filter(v.astype(np.float64), 15, mode="reflect", origin=(3, -6, 0))
        got = ndi.uniform_filter(vd, 15, mode="reflect", origin=(3, -6, 0)).get()
        assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max(), shape
        # forced z chunking (ramp planes at every chunk start) and the two-launch streaming route
        fused = ndi.gaussian_filter(vd, 2.0).get()
        for nch in (1, 3):
            lib.mi_debug_set_long_zchunks(nch)
            try:
                assert np.array_equal(ndi.gaussian_filter(vd, 2.0).get(), fused), (shape, nch)
            finally:
                lib.mi_debug_set_long_zchunks(0)
        lib.mi_debug_set_sep3d_long(1)
        try:
            streamed = ndi.gaussian_filter(vd, 2.0).get()
        finally:
            lib.mi_debug_set_sep3d_long(0)
        assert np.abs(streamed - fused).max() <= 1e-6 * np.abs(fused).max(), shape


def test_fused_float32_minmax(gpu, ndi):
    """minmax3d_f32.hip (cubic sizes 3..9 in ONE launch, LDS-DMA staging, v_min3 / v_max3 chains with a first-tap NaN
    fix-up): equal to scipy.ndimage on finite data for every index-mapping mode, partial tiles and origins, and equal
    to the two streaming launches (compare-select arithmetic) also on data with NaN / inf / signed zeros."""
    import scipy.ndimage as sndi
    from cupyimg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(99)
    for shape in [(40, 48, 256), (33, 21, 260), (70, 100, 512), (20, 16, 64), (9, 9, 16), (5, 70, 300), (130, 40, 252)]:
        v = rng.standard_normal(shape).astype(np.float32)
        vd = gpu.asarray(v)
        for mode in ["reflect", "mirror", "nearest", "wrap"]:
            for size in (3, 5, 7, 9):
                assert np.array_equal(ndi.minimum_filter(vd, size=size, mode=mode).get(), sndi.minimum_filter(v, size=size, mode=mode)), (shape, mode, size)
            assert np.array_equal(ndi.maximum_filter(vd, size=7, mode=mode).get(), sndi.maximum_filter(v, size=7, mode=mode)), (shape, mode)
        assert np.array_equal(ndi.grey_dilation(vd, size=5).get(), sndi.grey_dilation(v, size=5)), shape
        assert np.array_equal(ndi.minimum_filter(vd, size=5, origin=(1, -2, 0)).get(), sndi.minimum_filter(v, size=5, origin=(1, -2, 0))), shape
        # infinities and signed zeros: still SciPy's numbers
        w = v.copy()
        idx = rng.integers(0, w.size, size=max(4, w.size // 50))
        w.flat[idx[1::4]] = np.inf
        w.flat[idx[2::4]] = -np.inf
        w.flat[idx[3::4]] = -0.0
        wd = gpu.asarray(w)
        for fn, ref in ((ndi.minimum_filter, sndi.minimum_filter), (ndi.maximum_filter, sndi.maximum_filter)):
            assert np.array_equal(fn(wd, size=5, mode="mirror").get(), ref(w, size=5, mode="mirror")), (shape, ref.__name__)
        # NaNs: no path agrees with SciPy there (its 1-D min/max filter has its own NaN behaviour, and the
        # compare-select form of the reference depends on the pass order); what must hold: an output is NaN exactly
        # where the streaming launches give NaN (the window's first corner is NaN), and windows without a NaN are exact
        w.flat[idx[0::4]] = np.nan
        wd = gpu.asarray(w)
        clean = sndi.maximum_filter(np.isnan(w).astype(np.uint8), size=7, mode="mirror") == 0
        for fn, ref in ((ndi.minimum_filter, sndi.minimum_filter), (ndi.maximum_filter, sndi.maximum_filter)):
            fused = fn(wd, size=7, mode="mirror").get()
            lib.mi_debug_set_minmax_f32_fused(0)
            try:
                streamed = fn(wd, size=7, mode="mirror").get()
            finally:
                lib.mi_debug_set_minmax_f32_fused(1)
            assert np.array_equal(np.isnan(fused), np.isnan(streamed)), (shape, ref.__name__)
            assert np.array_equal(fused[clean], ref(np.where(np.isnan(w), np.float32(0), w), size=7, mode="mirror")[clean]), (shape, ref.__name__)


# ------------------------------------------------------------------ r3: order-1 constant-mode kernels (wide stores / loads)
def _c1_cases():
    rng = np.random.default_rng(151)
    ang = np.deg2rad(11.0)
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    return [
        # (input shape, matrix, offset, output shape): full 64 x 16 blocks, partial blocks, up / down scaling
        ((20, 37, 136), np.eye(3), np.zeros(3), None),                               # identity: every coordinate integral
        ((20, 37, 136), np.eye(3), np.array([1.0, -2.0, 3.0]), None),                # integer shift: exact boundary hits
        ((24, 40, 128), np.diag([1.02, 1.0, 1.0]) @ R, np.array([0.5, -1.25, 2.0]), None),
        ((24, 40, 128), np.diag([0.5, 0.75, 0.9]), np.array([0.25, 0.5, 0.125]), (30, 48, 192)),
        ((17, 33, 72), np.diag([(17 - 1) / (21 - 1.0), (33 - 1) / (47 - 1.0), (72 - 1) / (139 - 1.0)]), np.zeros(3), (21, 47, 140)),
        ((16, 16, 64), R @ np.diag([1.3, 0.8, 1.1]), rng.standard_normal(3) * 3, (18, 35, 68)),
    ]


@pytest.mark.parametrize("case", range(6))
def test_order1_constant_kernels_r3(gpu, ndi, case):
    """affine_transform / map_coordinates, order 1, mode constant, float32 volumes: the r3 kernels (16-byte stores and
    coordinate loads through an LDS tile, hoisted row prefix, v_fract_f64 split) against the oracle and, voxel for
    voxel, against the round-2 kernels they replace."""
    from cupyimg_amd import _lib
    lib = _lib.load()
    shape, M, off, oshape = _c1_cases()[case]
    rng = np.random.default_rng(152 + case)
    x = rng.standard_normal(shape).astype(np.float32)
    xd = gpu.asarray(x)
    oshape = shape if oshape is None else oshape
    ref = orc.affine_transform(x, M, off, output_shape=oshape, order=1, mode="constant", cval=-0.75)
    outs = {}
    for var in (1, 2, 3, 0):
        lib.mi_debug_set_interp_c1(var)
        try:
            outs[var] = ndi.affine_transform(xd, M, off, output_shape=oshape, order=1, mode="constant", cval=-0.75).get()
        finally:
            lib.mi_debug_set_interp_c1(1)
    assert np.allclose(outs[1], ref, rtol=0, atol=2e-6 * max(1.0, np.abs(ref).max())), np.abs(outs[1] - ref).max()
    assert np.array_equal(outs[1], outs[2]) and np.array_equal(outs[1], outs[3])
    # the round-2 kernel tests "inside" on the float32 weight; a double fraction < 2^-149 at the last sample is the
    # only way the two can differ, and none of these cases has one
    assert np.array_equal(outs[1], outs[0])
    # the same warp as explicit float32 coordinates
    idx = np.indices(oshape).reshape(3, -1).astype(np.float64)
    coords = (M @ idx + off[:, None]).reshape((3,) + tuple(oshape)).astype(np.float32)
    refm = orc.map_coordinates(x, coords, order=1, mode="constant", cval=-0.75)
    cd = gpu.asarray(coords)
    outm = {}
    for var in (1, 2, 3, 6, 7, 0):
        lib.mi_debug_set_interp_c1(var)
        try:
            outm[var] = ndi.map_coordinates(xd, cd, order=1, mode="constant", cval=-0.75).get()
        finally:
            lib.mi_debug_set_interp_c1(1)
    assert np.allclose(outm[1], refm, rtol=0, atol=2e-6 * max(1.0, np.abs(refm).max()))
    assert all(np.array_equal(outm[1], outm[v]) for v in (2, 3, 6, 7, 0))


def test_order1_constant_kernels_r3_nonfinite_and_edges(gpu, ndi):
    """inf / nan samples stay out of voxels whose skipped upper tap would touch them (the reference skips the second
    tap at integral coordinates, _interp_kernels.py:416; the oracle restates that); coordinates exactly on the last
    sample are inside, one ulp beyond is outside (cval)."""
    import scipy.ndimage as sndi
    x = np.random.default_rng(160).standard_normal((8, 16, 64)).astype(np.float32)
    xn = x.copy()
    xn[3, 5, 20] = np.inf
    xn[4, 7, 30] = np.nan
    xn[7, 15, 63] = -np.inf
    xd = gpu.asarray(xn)
    for off in ([0.0, 0.0, 0.0], [0.5, 0.0, 0.0], [0.0, 0.25, 0.0], [0.0, 0.0, 0.75], [1.0, 1.0, 1.0]):
        ref = orc.affine_transform(xn, np.eye(3), off, order=1, mode="constant", cval=2.0)
        got = ndi.affine_transform(xd, np.eye(3), off, order=1, mode="constant", cval=2.0).get()
        assert np.array_equal(np.isnan(got), np.isnan(ref)), off
        assert np.array_equal(np.isposinf(got), np.isposinf(ref)) and np.array_equal(np.isneginf(got), np.isneginf(ref)), off
        ok = np.isfinite(ref)
        assert np.allclose(got[ok], ref[ok], rtol=0, atol=2e-6 * np.abs(ref[ok]).max()), off
    # coordinates straddling the last sample of every axis (finite data, SciPy as the judge)
    c = np.zeros((3, 4, 16, 64), np.float32)
    c[0] = 7.0; c[1] = 15.0; c[2] = 63.0
    c[0, 1] = np.nextafter(np.float32(7.0), np.float32(8.0))
    c[1, 2] = np.nextafter(np.float32(15.0), np.float32(16.0))
    c[2, 3] = np.nextafter(np.float32(63.0), np.float32(64.0))
    ref = sndi.map_coordinates(x.astype(np.float64), c.astype(np.float64), order=1, mode="constant", cval=2.0)
    got = ndi.map_coordinates(gpu.asarray(x), gpu.asarray(c), order=1, mode="constant", cval=2.0).get()
    assert np.array_equal(got, ref.astype(np.float32))


# ------------------------------------------------------------------ r3: spline orders 2-5 on rank > 3 arrays
@pytest.mark.parametrize("order", [2, 3, 4, 5])
def test_spline_orders_on_rank4_and_rank5(gpu, ndi, order):
    """The reference's interpolation kernels are rank-generic (_interp_kernels.py:473-549); orders 2-5 used to stop at
    rank 3 here.  Against scipy.ndimage (double arithmetic on both sides: 1e-11; integer outputs exact)."""
    import scipy.ndimage as sndi
    rng = np.random.default_rng(170 + order)
    x4 = rng.standard_normal((5, 6, 7, 9))
    x5 = rng.standard_normal((3, 4, 5, 4, 6)).astype(np.float32)
    u4 = rng.integers(0, 200, size=(4, 5, 6, 8)).astype(np.uint8)
    for mode in ["constant", "nearest", "mirror", "reflect", "grid-wrap", "grid-constant", "wrap"]:
        # affine (incl. shift / zoom through it) on rank 4
        M = np.eye(4) + 0.05 * rng.standard_normal((4, 4))
        off = rng.standard_normal(4)
        ref = sndi.affine_transform(x4, M, off, order=order, mode=mode, cval=0.5)
        got = ndi.affine_transform(gpu.asarray(x4), M, off, order=order, mode=mode, cval=0.5).get()
        assert np.allclose(got, ref, rtol=0, atol=1e-11 * max(1.0, np.abs(ref).max())), (mode, "affine4")
        ref = sndi.shift(x4, [0.3, -1.2, 0.7, 2.1], order=order, mode=mode, cval=-1.0)
        got = ndi.shift(gpu.asarray(x4), [0.3, -1.2, 0.7, 2.1], order=order, mode=mode, cval=-1.0).get()
        assert np.allclose(got, ref, rtol=0, atol=1e-11 * max(1.0, np.abs(ref).max())), (mode, "shift4")
        # map_coordinates on rank 5, float32 data (SciPy and the device both filter in double)
        coords = np.stack([rng.uniform(-1.5, n + 0.5, size=(7, 11)) for n in x5.shape])
        ref = sndi.map_coordinates(x5.astype(np.float64), coords, order=order, mode=mode, cval=0.25)
        got = ndi.map_coordinates(gpu.asarray(x5), gpu.asarray(coords), order=order, mode=mode, cval=0.25, output=np.float64).get()
        assert np.allclose(got, ref, rtol=0, atol=1e-11 * max(1.0, np.abs(ref).max())), (mode, "map5")
    ref = sndi.zoom(u4, 1.3, order=order, mode="mirror")
    got = ndi.zoom(gpu.asarray(u4), 1.3, order=order, mode="mirror").get()
    assert got.dtype == np.uint8 and got.shape == ref.shape
    assert np.count_nonzero(got != ref) <= ref.size // 2000          # exact up to .5 ties decided by the last bit
    pre = sndi.spline_filter(x4, order=order, mode="mirror")
    assert np.allclose(ndi.spline_filter(gpu.asarray(x4), order=order, mode="mirror").get(), pre, rtol=0, atol=1e-11)


# ------------------------------------------------------------------ r3: rank filters without a size / rank limit
def test_rank_filters_beyond_128_taps_and_rank3(gpu, ndi):
    """median / rank / percentile filters with footprints of more than 128 samples and on rank-4 / rank-5 arrays (the
    scratch-column shell-sort kernel; the reference's shell-sort path has no limit either, filters.py:1753-1768,
    1829-1835).  Selection results are exact."""
    import scipy.ndimage as sndi
    rng = np.random.default_rng(180)
    x3 = rng.standard_normal((12, 14, 18)).astype(np.float32)
    for mode in ("reflect", "constant", "wrap"):
        ref = sndi.median_filter(x3, size=7, mode=mode, cval=0.5)                         # 343 samples
        assert np.array_equal(ndi.median_filter(gpu.asarray(x3), size=7, mode=mode, cval=0.5).get(), ref), mode
    img = rng.integers(0, 1 << 16, size=(40, 52)).astype(np.uint16)
    fp = rng.random((11, 13)) > 0.06                                                     # 143 positions, a few dropped
    fp[0, :] = True; fp[:, 0] = True
    assert 128 < fp.sum() < 143
    for rank in (0, 5, int(fp.sum()) // 2, int(fp.sum()) - 2):
        ref = sndi.rank_filter(img, rank, footprint=fp, mode="mirror", origin=(1, -2))
        assert np.array_equal(ndi.rank_filter(gpu.asarray(img), rank, footprint=fp, mode="mirror", origin=(1, -2)).get(), ref), rank
    x4 = rng.integers(-50, 50, size=(5, 6, 7, 8)).astype(np.int64) * (1 << 40) + rng.integers(0, 7, size=(5, 6, 7, 8))
    ref = sndi.median_filter(x4, size=(3, 3, 3, 3), mode="nearest")                      # rank 4, 81 samples, 64-bit exact
    assert np.array_equal(ndi.median_filter(gpu.asarray(x4), size=(3, 3, 3, 3), mode="nearest").get(), ref)
    x5 = rng.standard_normal((3, 4, 5, 4, 6))
    ref = sndi.percentile_filter(x5, 30, size=(1, 3, 3, 2, 3), mode="reflect")
    assert np.array_equal(ndi.percentile_filter(gpu.asarray(x5), 30, size=(1, 3, 3, 2, 3), mode="reflect").get(), ref)
    big = rng.standard_normal((40, 300, 300)).astype(np.float32)                          # more voxels than scratch columns
    ref = sndi.median_filter(big, size=(5, 6, 6), mode="reflect")                         # 180 samples, even sizes
    assert np.array_equal(ndi.median_filter(gpu.asarray(big), size=(5, 6, 6), mode="reflect").get(), ref)


# ------------------------------------------------------------------ r3: float16 images keep their dtype
def test_float16_images_keep_their_dtype(gpu, ndi):
    """The reference filters a float16 image into a float16 result (_filters_core.py:169-171, arithmetic in float32 /
    float64); round 2 silently returned float32.  Here: converted to float32 on the device (exact), filtered, rounded
    to float16 -- compared with SciPy on the exactly-converted float32 image, rounded the same way."""
    import scipy.ndimage as sndi
    rng = np.random.default_rng(190)
    h = rng.standard_normal((24, 40, 64)).astype(np.float16)
    hd = gpu.asarray(h)
    assert hd.dtype == np.float16 and np.array_equal(hd.get(), h)
    f = h.astype(np.float32)
    cases = [
        (lambda a, **k: ndi.uniform_filter(a, 5, **k), lambda a: sndi.uniform_filter(a, 5), 2),
        (lambda a, **k: ndi.gaussian_filter(a, 1.5, **k), lambda a: sndi.gaussian_filter(a, 1.5), 2),
        (lambda a, **k: ndi.correlate1d(a, [1, 2, 1], axis=1, **k), lambda a: sndi.correlate1d(a, [1, 2, 1], axis=1), 2),
        (lambda a, **k: ndi.grey_erosion(a, size=3, **k), lambda a: sndi.grey_erosion(a, size=3), 0),
        (lambda a, **k: ndi.median_filter(a, size=3, **k), lambda a: sndi.median_filter(a, size=3), 0),
        (lambda a, **k: ndi.affine_transform(a, np.eye(3) * 0.9, order=1, **k), lambda a: sndi.affine_transform(a, np.eye(3) * 0.9, order=1), 2),
    ]
    for got_fn, ref_fn, ulps in cases:
        got = got_fn(hd)
        assert got.dtype == np.float16
        ref = ref_fn(f.astype(np.float64)).astype(np.float16)
        g = got.get()
        if ulps == 0:
            assert np.array_equal(g, ref)
        else:
            # float32 arithmetic rounded to float16 against float64 arithmetic rounded to float16: at most one ulp
            assert np.abs(g.view(np.int16).astype(np.int32) - ref.view(np.int16).astype(np.int32)).max() <= 1
        out16 = gpu.empty(h.shape, np.float16)
        assert got_fn(hd, output=out16) is out16 and np.array_equal(out16.get(), g)
        assert got_fn(gpu.asarray(f), output=np.float16).dtype == np.float16           # float16 asked for explicitly
        assert got_fn(hd, output=np.float32).dtype == np.float32                        # ... or not
    assert ndi.uniform_filter(h, 3).dtype == np.float16                                # host float16 arrays as well
    b = ndi.binary_erosion(hd)                                                          # float16 as a mask source
    assert b.dtype == np.bool_ and np.array_equal(b.get(), sndi.binary_erosion(f))


# ------------------------------------------------------------------ r3: affine_transform with the gathers out of LDS
def test_affine_lds_staged_kernel(gpu, ndi):
    """Outputs of >= 2^18 voxels take the LDS-staged kernel whenever the bounding box of a 64 x 8 x 8 output tile fits
    its LDS budget (decided from the matrix): results bit-identical to the L1-gather kernel (knob 5) for identity,
    integer shifts (exact boundary hits), shears / small rotations about every axis, zooms (in and out, the latter falls
    back when the box is too large), partial tiles, inputs smaller than a box, non-finite samples; and within 2e-6 of
    the oracle."""
    from cupyimg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(200)

    def rot(axis, deg):
        a = np.deg2rad(deg); c, s = np.cos(a), np.sin(a)
        R = np.eye(3); i, j = [(1, 2), (0, 2), (0, 1)][axis]
        R[i, i] = c; R[i, j] = -s; R[j, i] = s; R[j, j] = c
        return R

    cases = [
        ((64, 64, 64), np.eye(3), np.zeros(3), None),
        ((64, 64, 64), np.eye(3), np.array([1.0, -2.0, 3.0]), None),
        ((40, 72, 136), np.diag([1.02, 1.0, 1.0]) @ rot(0, 7), np.array([0.5, -1.25, 2.0]), (66, 70, 132)),
        ((48, 80, 96), rot(1, 5) @ rot(2, -4), rng.standard_normal(3) * 2, (70, 75, 128)),
        ((50, 60, 70), np.diag([0.6, 0.7, 0.8]), np.array([0.3, 0.2, 0.1]), (80, 84, 88)),      # up-sampling: small boxes
        ((90, 90, 200), np.diag([1.6, 1.3, 1.5]), np.zeros(3), (56, 69, 132)),                   # down-sampling
        ((64, 64, 64), rot(2, 30), np.array([10.0, -5.0, 0.0]), (64, 64, 72)),                   # box too large: falls back
        ((3, 5, 8), np.diag([0.04, 0.07, 0.05]), np.zeros(3), (64, 64, 64)),                     # input smaller than a box
        ((64, 64, 64), -np.eye(3), np.array([63.0, 63.0, 63.0]), None),                          # flip: coordinates hit 0 and n-1 exactly
    ]
    for shape, M, off, oshape in cases:
        x = rng.standard_normal(shape).astype(np.float32)
        if shape == (64, 64, 64):
            x[10, 20, 30] = np.inf; x[11, 21, 31] = np.nan
        xd = gpu.asarray(x)
        oshape = shape if oshape is None else oshape
        outs = {}
        lib.mi_debug_set_affine_zstream(0)             # r4: matrices that leave axis 0 to itself would stream along z instead
        try:
            for var in (1, 5):
                lib.mi_debug_set_interp_c1(var)
                try:
                    outs[var] = ndi.affine_transform(xd, M, off, output_shape=oshape, order=1, mode="constant", cval=-0.75).get()
                finally:
                    lib.mi_debug_set_interp_c1(1)
        finally:
            lib.mi_debug_set_affine_zstream(1)
        assert np.array_equal(outs[1], outs[5], equal_nan=True), (shape, np.abs(outs[1] - outs[5]).max())
        zs = ndi.affine_transform(xd, M, off, output_shape=oshape, order=1, mode="constant", cval=-0.75).get()
        assert np.array_equal(zs, outs[5], equal_nan=True), (shape, "default dispatch")
        lib.mi_debug_set_affine_zstream(0)
        # workgroups that walk several tiles of a column (next origin computed one tile ahead): same voxels
        for gz in (1, 3):
            lib.mi_debug_set_affine_gz(gz)
            try:
                walked = ndi.affine_transform(xd, M, off, output_shape=oshape, order=1, mode="constant", cval=-0.75).get()
            finally:
                lib.mi_debug_set_affine_gz(0)
            assert np.array_equal(walked, outs[1], equal_nan=True), (shape, gz)
        lib.mi_debug_set_affine_zstream(1)
        ref = orc.affine_transform(x, M, off, output_shape=oshape, order=1, mode="constant", cval=-0.75)
        ok = np.isfinite(ref)
        assert np.array_equal(np.isfinite(outs[1]), ok), shape
        assert np.allclose(outs[1][ok], ref[ok], rtol=0, atol=2e-6 * max(1.0, np.abs(ref[ok]).max())), shape
        # the same warp as explicit float32 coordinates: map_coordinates finds the box of every tile itself
        idx = np.indices(oshape).reshape(3, -1).astype(np.float64)
        coords = (M @ idx + off[:, None]).reshape((3,) + tuple(oshape)).astype(np.float32)
        cd = gpu.asarray(coords)
        outm = {}
        for var in (4, 7, 1, 6):
            lib.mi_debug_set_interp_c1(var)
            try:
                outm[var] = ndi.map_coordinates(xd, cd, order=1, mode="constant", cval=-0.75).get()
            finally:
                lib.mi_debug_set_interp_c1(1)
        assert all(np.array_equal(outm[v], outm[6], equal_nan=True) for v in (4, 7, 1)), (shape, "map_coordinates")
    # (knob 4 = the LDS-staged map_coordinates kernel, not the default: slower on config D, see interp_fast.hip)
    # coordinates with no structure at all (every workgroup's box is the whole volume: the L1 path inside the LDS kernel),
    # smooth ones with wild outliers, NaN / inf coordinates
    x = rng.standard_normal((40, 50, 60)).astype(np.float32)
    xd = gpu.asarray(x)
    oshape = (64, 64, 96)
    wild = np.stack([rng.uniform(-3, n + 2, size=oshape) for n in x.shape]).astype(np.float32)
    smooth = np.stack(np.meshgrid(*[np.linspace(-1, n, m) for n, m in zip(x.shape, oshape)], indexing="ij")).astype(np.float32)
    smooth += 0.3 * rng.standard_normal(smooth.shape).astype(np.float32)
    spiky = smooth.copy()
    spiky[:, ::7, ::5, ::11] = wild[:, ::7, ::5, ::11]
    spiky[0, 3, 4, 5] = np.nan; spiky[1, 8, 9, 10] = np.inf; spiky[2, 20, 21, 22] = -np.inf
    for coords in (wild, smooth, spiky):
        cd = gpu.asarray(coords)
        outm = {}
        for var in (4, 7, 1, 6):
            lib.mi_debug_set_interp_c1(var)
            try:
                outm[var] = ndi.map_coordinates(xd, cd, order=1, mode="constant", cval=1.5).get()
            finally:
                lib.mi_debug_set_interp_c1(1)
        assert all(np.array_equal(outm[v], outm[6], equal_nan=True) for v in (4, 7, 1))
        ok = np.isfinite(coords).all(axis=0)
        ref = orc.map_coordinates(x, np.where(np.isfinite(coords), coords, -5.0), order=1, mode="constant", cval=1.5)
        assert np.allclose(outm[1][ok], ref[ok], rtol=0, atol=2e-6 * max(1.0, np.abs(ref).max()))


def test_long_kernel_generations_agree(gpu, ndi):
    """sep3d_long3_kernel (the product kernel: y pass one plane ahead of the x / z passes, x pass as op_sel packed FMAs)
    against the r2 instruction stream kept behind mi_debug_set_long_rows(1) for 9 / 13 / 17 taps: same voxels within the
    reordering of one float32 FMA chain, both against the oracle; every boundary mode, partial tiles, anisotropic
    weights (the re-loading variant), origins."""
    from cupyimg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(210)
    for shape in [(40, 37, 64), (33, 21, 264), (19, 50, 256)]:
        x = rng.standard_normal(shape).astype(np.float32)
        xd = gpu.asarray(x)
        for mode in MODES:
            for size in (9, 13, 17):
                try:
                    lib.mi_debug_set_long_rows(1)
                    a = ndi.uniform_filter(xd, size, mode=mode, cval=0.75).get()
                    lib.mi_debug_set_long_rows(0)
                    b = ndi.uniform_filter(xd, size, mode=mode, cval=0.75).get()
                finally:
                    lib.mi_debug_set_long_rows(0)
                ref = orc.uniform_filter(x, size, mode=mode, cval=0.75)
                assert maxnorm_rel(a, ref) <= 1e-6, (shape, mode, size)
                assert maxnorm_rel(b, ref) <= 1e-6, (shape, mode, size)
                assert maxnorm_rel(a, b) <= 1e-6, (shape, mode, size)
            g = ndi.gaussian_filter(xd, [2.0, 1.6, 1.9], mode=mode, cval=-0.5).get()
            assert maxnorm_rel(g, orc.gaussian_filter(x, [2.0, 1.6, 1.9], mode=mode, cval=-0.5)) <= 1e-6, (shape, mode)
            # equal tap counts with DIFFERENT weights per axis: the variants that re-load their weights every step
            # (17, 9, 7 and 5 taps; the last two take the long kernel behind the knob at these sizes)
            for sig in ([2.0, 1.9, 1.95], [1.0, 1.1, 1.05], [0.75, 0.8, 0.7], [0.5, 0.6, 0.55]):
                try:
                    lib.mi_debug_set_sep3d_long(2)
                    g = ndi.gaussian_filter(xd, sig, mode=mode, cval=-0.5).get()
                finally:
                    lib.mi_debug_set_sep3d_long(0)
                assert maxnorm_rel(g, orc.gaussian_filter(x, sig, mode=mode, cval=-0.5)) <= 1e-6, (shape, mode, sig)
            # 3 / 5 / 7 taps through the long kernel (the default on chip-filling volumes), incl. origins
            for size in (3, 5, 7):
                try:
                    lib.mi_debug_set_sep3d_long(2)
                    b = ndi.uniform_filter(xd, size, mode=mode, cval=0.75).get()
                    c = ndi.uniform_filter(xd, size, mode=mode, cval=0.75, origin=[1, -1, 0]).get()
                finally:
                    lib.mi_debug_set_sep3d_long(0)
                assert maxnorm_rel(b, orc.uniform_filter(x, size, mode=mode, cval=0.75)) <= 1e-6, (shape, mode, size)
                assert maxnorm_rel(c, orc.uniform_filter(x, size, mode=mode, cval=0.75, origin=[1, -1, 0])) <= 1e-6, (shape, mode, size)
            u = ndi.uniform_filter(xd, 11, mode=mode, origin=[2, -3, 0]).get()
            assert maxnorm_rel(u, orc.uniform_filter(x, 11, mode=mode, origin=[2, -3, 0])) <= 1e-6, (shape, mode)


def test_long_kernel_anisotropic_tap_pairs(gpu, ndi):
    """Volumes with anisotropic voxels: fewer (r4: or more) taps along z than in the plane take ONE launch of the long kernel
    (sep3d_long3_kernel<W, false, false, 0, WZ>) for the (W, WZ) pairs it is built for, on volumes of >= 4 Mvoxels;
    every index-mapping mode, gaussian and uniform weights, an origin along z; `constant` mode and other pairs fall back
    to the streaming passes and must agree as well."""
    import cupyimg_amd as ca
    rng = np.random.default_rng(77)
    shape = (64, 256, 256)
    x = rng.standard_normal(shape).astype(np.float32)
    xd = gpu.asarray(x)
    for sig, taps in (([1.0, 2.0, 2.0], (17, 9)), ([0.5, 1.0, 1.0], (9, 5)), ([0.75, 1.5, 1.5], (13, 7)), ([0.5, 2.0, 2.0], (17, 5)),
                      ([0.25, 0.5, 0.5], (5, 3)), ([0.625, 1.25, 1.25], (11, 7)), ([0.875, 1.75, 1.75], (15, 9)), ([0.375, 0.75, 0.75], (7, 5)),
                      # r4: MORE taps along z than in the plane
                      ([2.0, 1.0, 1.0], (9, 17)), ([1.5, 1.0, 1.0], (9, 13)), ([2.0, 1.5, 1.5], (13, 17)), ([1.0, 0.5, 0.5], (5, 9)),
                      ([1.5, 0.5, 0.5], (5, 13)), ([2.0, 0.5, 0.5], (5, 17)), ([1.5, 0.75, 0.75], (7, 13))):
        for mode in ("reflect", "mirror", "nearest", "wrap"):
            g = ndi.gaussian_filter(xd, sig, mode=mode).get()
            assert "sep3d_long3_kernel<%d,false,false,0,%d>" % taps in ca.last_kernel(), (sig, ca.last_kernel())
            assert maxnorm_rel(g, orc.gaussian_filter(x, sig, mode=mode)) <= 1e-6, (sig, mode)
        g = ndi.gaussian_filter(xd, sig, mode="constant", cval=0.5).get()           # falls back
        assert "sep3d_long3_kernel" not in ca.last_kernel()
        assert maxnorm_rel(g, orc.gaussian_filter(x, sig, mode="constant", cval=0.5)) <= 1e-6, sig
    for size, taps in (((3, 9, 9), (9, 3)), ((13, 17, 17), (17, 13)), ((7, 9, 9), (9, 7)), ((17, 9, 9), (9, 17)), ((13, 7, 7), (7, 13))):
        u = ndi.uniform_filter(xd, size, mode="reflect").get()
        assert "sep3d_long3_kernel<%d,false,false,0,%d>" % taps in ca.last_kernel(), (size, ca.last_kernel())
        assert maxnorm_rel(u, orc.uniform_filter(x, size, mode="reflect")) <= 1e-6, size
        u = ndi.uniform_filter(xd, size, mode="mirror", origin=[1, -2, 0]).get()
        assert maxnorm_rel(u, orc.uniform_filter(x, size, mode="mirror", origin=[1, -2, 0])) <= 1e-6, size
    u = ndi.uniform_filter(xd, (11, 17, 17)).get()                                  # not a built pair: streaming passes
    assert "sep3d_long3_kernel" not in ca.last_kernel()
    assert maxnorm_rel(u, orc.uniform_filter(x, (11, 17, 17))) <= 1e-6


def test_rank_filter_int64_beyond_2p53_follows_scipy(gpu, ndi):
    """SciPy's NI_RankFilter / NI_MinOrMaxFilter hold the window in doubles, so 64-bit integers beyond 2^53 come back
    rounded; both rank kernels (registers: rank <= 3 arrays; scratch columns: rank 4 arrays / large footprints) and the
    min / max kernels reproduce that bit for bit instead of returning the unrounded element."""
    import scipy.ndimage as sndi
    x = (np.arange(5 * 8 * 9, dtype=np.int64).reshape(5, 8, 9) * 3 + (1 << 60) + 1)
    np.random.default_rng(5).shuffle(x.reshape(-1))
    for dt in (np.int64, np.uint64):
        v = x.astype(dt)
        assert np.array_equal(ndi.median_filter(gpu.asarray(v), size=3).get(), sndi.median_filter(v, size=3))
        assert np.array_equal(ndi.minimum_filter(gpu.asarray(v), size=3).get(), sndi.minimum_filter(v, size=3))
        v4 = v.reshape(1, 5, 8, 9)
        assert np.array_equal(ndi.rank_filter(gpu.asarray(v4), rank=5, size=(1, 3, 3, 3)).get(),
                              sndi.rank_filter(v4, rank=5, size=(1, 3, 3, 3)))
        assert np.array_equal(ndi.median_filter(gpu.asarray(v), size=(5, 5, 7)).get(), sndi.median_filter(v, size=(5, 5, 7)))


# ------------------------------------------------------------------ r4: affine_transform streaming along z (axis 0 decoupled)
def test_affine_zstream_kernel(gpu, ndi):
    """Matrices that leave axis 0 (or axis 1: the second block of cases) to itself (in-plane rotation / shear / scaling + a scaling / shift through the slices:
    BASELINE config D') stream along z (affine3d_zstream_kernel: in-plane addresses and weights once per workgroup, input
    planes through a ring of four LDS slots).  Bit-identical to the L1-gather kernel (knob 5) for both tile heights and
    several z chunkings; rotations up to 45 deg, slice steps of 0 / 0.5 / 1.02 / 1.7 / 2 / -1 (2.5 falls back), shifts that
    push part of the output outside (cval), partial tiles, output shapes unlike the input's, non-finite samples; within
    2e-6 of the oracle."""
    from cupyimg_amd import _lib, last_kernel
    lib = _lib.load()
    rng = np.random.default_rng(400)

    def inplane(deg, sy=1.0, sx=1.0, shear=0.0):
        a = np.deg2rad(deg); c, s = np.cos(a), np.sin(a)
        return np.array([[c * sy, -s * sx + shear], [s * sy, c * sx]])

    def mat(m0, A):
        M = np.zeros((3, 3)); M[0, 0] = m0; M[1:, 1:] = A
        return M

    cases = [
        ((40, 72, 136), mat(1.02, inplane(7)), np.array([0.5, -1.25, 2.0]), (66, 70, 132), True),
        ((64, 64, 64), mat(1.0, np.eye(2)), np.array([0.0, 0.0, 0.0]), None, True),
        ((64, 64, 64), mat(1.0, np.eye(2)), np.array([3.0, -2.0, 5.0]), None, True),                 # integer shifts: exact boundary hits
        ((33, 100, 130), mat(0.5, inplane(-20)), np.array([2.0, 40.0, -30.0]), (70, 96, 128), True),      # (until r4b the 64-row tiles' bounding rectangle was wider than the pitch: the window is sheared now)
        ((80, 90, 100), mat(1.7, inplane(45)), np.array([-3.0, 60.0, -20.0]), (50, 128, 128), (1, 32)),   # 64-row tiles: the window does not fit LDS
        ((80, 64, 200), mat(2.0, inplane(0, 0.8, 1.1)), np.array([0.25, 1.0, 2.0]), (40, 80, 164), True),
        ((64, 64, 64), mat(-1.0, inplane(180)), np.array([63.0, 63.0, 63.0]), None, True),            # flips: coordinates hit 0 and n - 1
        ((20, 70, 90), mat(0.0, inplane(3, shear=0.1)), np.array([7.3, 1.0, -4.0]), (64, 64, 64), True),   # every output plane samples z = 7.3
        ((90, 64, 64), mat(2.5, inplane(5)), np.zeros(3), (36, 64, 128), False),                       # |m00| > 2: box / gather kernels
        ((64, 64, 64), mat(1.0, inplane(0, 1.3, 1.3)), np.zeros(3), (64, 64, 64), False),              # rectangle wider than the pitch
        ((3, 5, 8), mat(0.04, np.diag([0.07, 0.05])), np.zeros(3), (64, 64, 64), True),                # input smaller than a rectangle
    ]
    # the same with axis 1 as the axis the matrix leaves to itself (rotations in the (z, x) plane): the kernel streams
    # along y through (z, x) slices
    def mat1(m1, A):
        M = np.zeros((3, 3)); M[1, 1] = m1
        M[0, 0], M[0, 2], M[2, 0], M[2, 2] = A[0, 0], A[0, 1], A[1, 0], A[1, 1]
        return M
    cases += [
        ((72, 40, 136), mat1(1.02, inplane(7)), np.array([-1.25, 0.5, 2.0]), (70, 66, 132), True),
        ((100, 33, 130), mat1(0.5, inplane(-20)), np.array([40.0, 2.0, -30.0]), (96, 70, 128), True),
        ((64, 64, 64), mat1(-1.0, inplane(180)), np.array([63.0, 63.0, 63.0]), None, True),
        ((64, 80, 200), mat1(2.0, inplane(0, 0.8, 1.1)), np.array([1.0, 0.25, 2.0]), (80, 40, 164), True),
        ((64, 90, 64), mat1(2.5, inplane(5)), np.zeros(3), (64, 36, 128), False),
    ]
    for shape, M, off, oshape, takes in cases:
        x = rng.standard_normal(shape).astype(np.float32)
        if min(shape) > 12:
            x[10, 9, 11] = np.inf; x[11, 12, 10] = np.nan
        xd = gpu.asarray(x)
        oshape = shape if oshape is None else oshape
        lib.mi_debug_set_interp_c1(5)
        try:
            want = ndi.affine_transform(xd, M, off, output_shape=oshape, order=1, mode="constant", cval=-0.75).get()
        finally:
            lib.mi_debug_set_interp_c1(1)
        for ty in (1, 32, 64):
            for zc in (0, 1, 3):
                lib.mi_debug_set_affine_zstream(ty)
                lib.mi_debug_set_affine_zchunks(zc)
                try:
                    got = ndi.affine_transform(xd, M, off, output_shape=oshape, order=1, mode="constant", cval=-0.75).get()
                    expect = takes if isinstance(takes, bool) else ty in takes
                    assert ("zstream" in last_kernel() or "zrect" in last_kernel()) == expect, (shape, ty, last_kernel())   # rectangle / sheared form
                finally:
                    lib.mi_debug_set_affine_zstream(1)
                    lib.mi_debug_set_affine_zchunks(0)
                assert np.array_equal(got, want, equal_nan=True), (shape, ty, zc, float(np.nanmax(np.abs(got - want))))
        ref = orc.affine_transform(x, M, off, output_shape=oshape, order=1, mode="constant", cval=-0.75)
        ok = np.isfinite(ref)
        assert np.array_equal(np.isfinite(want), ok), shape
        assert np.allclose(want[ok], ref[ok], rtol=0, atol=2e-6 * max(1.0, np.abs(ref[ok]).max())), shape


# ------------------------------------------------------------------ r4: map_coordinates streaming along z (taps out of LDS)
def test_map_coordinates_zstream_kernel(gpu, ndi):
    """Outputs of >= 2^18 voxels with >= 64 columns take map_coords3d_zstream_kernel (order 1, `constant`, float32): a
    workgroup walks down its chunk of output planes, the bounding box of every plane's taps is reduced from the
    coordinates one plane ahead, the input planes come through a ring of four LDS slots with one shared rectangle.
    Bit-identical to the L1-gather kernel (knob 0) for smooth warps (rotation, zoom in / out, drift that re-centres the
    rectangle), for warps whose planes do not fit (every step falls back to the L1 gathers: tall rotations, z-mixing,
    random coordinates), coordinates outside the volume on every side, NaN / inf coordinates, partial tiles, several z
    chunkings; knob 2 forces the fallback path for every step."""
    from cupyimg_amd import _lib, last_kernel
    lib = _lib.load()
    rng = np.random.default_rng(500)
    shape = (40, 90, 150)
    x = rng.standard_normal(shape).astype(np.float32)
    x[7, 8, 9] = np.inf; x[11, 12, 13] = np.nan
    xd = gpu.asarray(x)
    oshape = (48, 70, 132)                                        # 132 = 2 x 64 + 4: partial tile; 70 = 2 x 32 + 6
    idx = np.indices(oshape).reshape(3, -1).astype(np.float64)

    def warp(M, off, extra=None):
        c = (np.asarray(M) @ idx + np.asarray(off, dtype=np.float64)[:, None]).reshape((3,) + oshape)
        if extra is not None:
            c = c + extra(c)
        return c.astype(np.float32)

    ang = np.deg2rad(7.0); cs, sn = np.cos(ang), np.sin(ang)
    cases = {
        "rotation in the plane": warp([[1.02, 0, 0], [0, cs, -sn], [0, sn, cs]], [0.5, 3.0, -2.0]),
        "identity, integer shift (exact boundary hits)": warp(np.eye(3), [2.0, -3.0, 5.0]),
        "zoom in": warp(np.diag([0.6, 0.7, 0.8]), [0.3, 0.2, 0.1]),
        "zoom out (rectangle too wide: L1 path)": warp(np.diag([0.8, 1.6, 1.7]), [0.0, 0.0, 0.0]),
        "z mixes with y (planes drift: re-centring, late fetches)": warp([[1.0, 0.12, 0.0], [-0.12, 1.0, 0.0], [0, 0, 1.0]], [-2.0, 6.0, 1.0]),
        "steep z mixing (more than four planes per step: L1 path)": warp([[0.9, 0.5, 0.3], [0.1, 1.0, 0.0], [0, 0, 1.0]], [0.0, 0.0, 0.0]),
        "smooth non-affine": warp(np.eye(3), [1.0, 2.0, 3.0], lambda c: 1.5 * np.sin(c / 9.0)),
        "mostly outside": warp(np.eye(3), [30.0, -60.0, 100.0]),
        "random coordinates": (rng.random((3,) + oshape) * np.array(shape)[:, None, None, None]).astype(np.float32),
        # r4b: the rectangle is placed from the tile's corner voxels; voxels whose taps it does not hold gather for themselves
        "bulge inside the tiles (corner box too small)": warp(np.eye(3), [1.0, 2.0, 3.0],
                                                             lambda c: np.stack([3.0 * np.sin(c[1] / 5.0) * np.sin(c[2] / 7.0),
                                                                                 14.0 * np.sin(c[2] / 10.0), 18.0 * np.sin(c[1] / 5.0)])),
        "jitter of +-12 voxels around the identity": warp(np.eye(3), [0.0, 0.0, 0.0], lambda c: rng.uniform(-12, 12, c.shape)),
        "jitter along z only": warp(np.eye(3), [0.0, 0.0, 0.0], lambda c: np.stack([rng.uniform(-3, 3, c.shape[1:]), 0 * c[1], 0 * c[2]])),
    }
    wild = cases["rotation in the plane"].copy()
    wild[0, 5, 6, 7] = np.nan; wild[1, 9, 10, 11] = np.inf; wild[2, 20, 21, 22] = -np.inf; wild[:, 30, 40, 50] = 1e30
    wild[0, 33] = -5.0                                            # a whole plane of coordinates outside
    cases["NaN / inf / huge coordinates"] = wild
    for name, c in cases.items():
        cd = gpu.asarray(c)
        lib.mi_debug_set_map_zstream(0)
        try:
            want = ndi.map_coordinates(xd, cd, order=1, mode="constant", cval=-0.75).get()
        finally:
            lib.mi_debug_set_map_zstream(1)
        # knob 3 = the exact box reduction (first r4 kernel); variant = 10 x voxels per thread + planes of coordinates in flight
        for knob, zc, variant in ((1, 0, 0), (1, 1, 0), (1, 5, 0), (2, 0, 0), (3, 0, 0), (3, 5, 0), (1, 0, 81), (1, 5, 81), (1, 0, 82),
                                  (1, 1, 82), (1, 5, 82), (1, 24, 82), (1, 24, 41), (2, 0, 82), (2, 0, 41)):
            lib.mi_debug_set_map_zstream(knob); lib.mi_debug_set_map_zchunks(zc); lib.mi_debug_set_map_zvariant(variant)
            try:
                got = ndi.map_coordinates(xd, cd, order=1, mode="constant", cval=-0.75).get()
                assert "map_coords3d_zstream" in last_kernel()
            finally:
                lib.mi_debug_set_map_zstream(1); lib.mi_debug_set_map_zchunks(0); lib.mi_debug_set_map_zvariant(0)
            assert np.array_equal(got, want, equal_nan=True), (name, knob, zc, variant, int(np.sum(~((got == want) | (np.isnan(got) & np.isnan(want))))))
        ref = orc.map_coordinates(x, c, order=1, mode="constant", cval=-0.75)
        ok = np.isfinite(ref)
        assert np.array_equal(np.isfinite(want), ok), name
        assert np.allclose(want[ok], ref[ok], rtol=0, atol=2e-6 * max(1.0, np.abs(ref[ok]).max())), name


def test_separable_filters_keep_nonfinite_samples_inside_their_window(gpu, ndi):
    """An inf / NaN sample makes exactly the outputs whose taps reach it non-finite -- what an explicit sum over the taps (the
    reference's kernels; SciPy's correlate1d) gives.  r4b: the packed x passes of the fused long kernels and of the
    streaming passes multiplied the sample next to a window with the zero that pads a weight pair (0 x inf = NaN one to
    three voxels beyond the taps along x); the matrix-core y pass of the experimental r4 kernel multiplies whole tile
    columns with the zeros of its band and recomputes a block tap by tap when that shows.  Reference: correlate1d per axis
    in float64 (SciPy's uniform_filter itself keeps a RUNNING sum and turns the rest of a line into NaN after an inf)."""
    import scipy.ndimage as sndi
    from cupyimg_amd import _lib, last_kernel
    lib = _lib.load()
    rng = np.random.default_rng(77)

    def explicit(v, weights, mode, cval=0.0):
        for ax, w in enumerate(weights):
            v = sndi.correlate1d(v, w, axis=ax, mode=mode, cval=cval)
        return v

    def gauss_w(sigma):
        r = int(4.0 * sigma + 0.5)
        xs = np.arange(-r, r + 1)
        w = np.exp(-0.5 * (xs / sigma) ** 2)
        return w / w.sum()

    seen = set()
    for shape in [(40, 64, 256), (36, 50, 300), (24, 128, 1024)]:
        x = rng.standard_normal(shape).astype(np.float32)
        x[shape[0] // 4, shape[1] // 3, shape[2] // 5] = np.inf
        x[shape[0] // 2, shape[1] // 2, (4 * shape[2]) // 5] = np.nan
        x[-1, -1, -1] = -np.inf
        x[1, 2, 3] = np.inf
        xd = gpu.asarray(x)
        # ("constant", a fill value): r5 -- a zero fill value runs on the r3 long kernel, any other on the r2 kernel with its correction
        cases = [("uniform", s, m) for s in (3, 5, 7, 9, 11, 13, 17, 21, 4) for m in ("reflect", "constant", ("constant", 0.75))]
        cases += [("gaussian", s, "reflect") for s in (1.0, 1.5, 2.0, 2.6, (2, 1, 1), (1, 2, 1.5))]
        for kind, par, mode in cases:
            cv = 0.0
            if isinstance(mode, tuple):
                mode, cv = mode
            for rows in ((0, 4) if (kind == "gaussian" and par in (1.0, 1.5, 2.0)) or (kind == "uniform" and par in (9, 13, 17) and mode == "reflect") else (0,)):
                lib.mi_debug_set_long_rows(rows)
                try:
                    if kind == "uniform":
                        got = ndi.uniform_filter(xd, par, mode=mode, cval=cv).get()
                        ws = [np.ones(par) / par] * 3
                    else:
                        got = ndi.gaussian_filter(xd, par, mode=mode, cval=cv).get()
                        sig = par if isinstance(par, tuple) else (par,) * 3
                        ws = [gauss_w(s) for s in sig]
                finally:
                    lib.mi_debug_set_long_rows(0)
                seen.add(last_kernel().split("<")[0].replace("mi::", ""))
                with np.errstate(all="ignore"):
                    ref = explicit(x.astype(np.float64), ws, mode, cv)
                what = (shape, kind, par, mode, cv, rows, last_kernel()[:48])
                assert np.array_equal(np.isnan(got), np.isnan(ref)), what
                assert np.array_equal(np.isposinf(got), np.isposinf(ref)) and np.array_equal(np.isneginf(got), np.isneginf(ref)), what
                fin = np.isfinite(ref)
                assert np.abs(got[fin] - ref[fin]).max() <= 1e-6 * np.abs(ref[fin]).max(), what
    assert {"sep3d_long3_kernel", "sep3d_long4_kernel", "sep3d_long_kernel", "sep3d_lean_kernel", "stream_pass_kernel"} <= seen, seen


def test_separable_filters_rows_not_a_multiple_of_four(gpu, ndi):
    """Volumes / images whose rows are not a multiple of four floats (181 x 217 x 181, 91 x 109 x 91, ...) take the fused
    kernels through an explicit extension of the rows along x (r4b: `_fused_3d_padded_rows`): every boundary mode, box and
    gaussian weights, short and long kernels, origins along y / z, a cval; against SciPy at the filters' tolerance."""
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(321)
    fused = ("sep3d_", "stream_pass_kernel", "box")
    for shape in [(45, 54, 45), (33, 40, 101), (20, 37, 262), (1, 301, 403), (16, 16, 31)]:
        x = rng.standard_normal(shape).astype(np.float32)
        xd = gpu.asarray(x)
        for mode in ("reflect", "mirror", "nearest", "wrap", "constant"):
            for what in (("u", 3), ("u", 5), ("u", 9), ("u", (3, 5, 7)), ("g", 1.0), ("g", 2.0), ("g", (0.0, 1.5, 1.5))):
                kw = dict(mode=mode, cval=1.25)
                if what[0] == "u":
                    got = ndi.uniform_filter(xd, what[1], **kw).get()
                    ref = sndi.uniform_filter(x.astype(np.float64), what[1], **kw)
                else:
                    got = ndi.gaussian_filter(xd, what[1], **kw).get()
                    ref = sndi.gaussian_filter(x.astype(np.float64), what[1], **kw)
                assert any(f in last_kernel() for f in fused), (shape, mode, what, last_kernel())
                assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (shape, mode, what, last_kernel())
        # origins along z / y (x origins are not taken by the fused kernels), output into a view
        got = ndi.uniform_filter(xd, 5, origin=(1 if shape[0] > 4 else 0, -1, 0)).get()
        ref = sndi.uniform_filter(x.astype(np.float64), 5, origin=(1 if shape[0] > 4 else 0, -1, 0))
        assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), shape
    img = rng.standard_normal((301, 403)).astype(np.float32)
    got = ndi.gaussian_filter(gpu.asarray(img), 1.5).get()
    assert np.abs(got - sndi.gaussian_filter(img.astype(np.float64), 1.5)).max() <= 1e-6 * 4
    # flat min / max and grey morphology: float32, uint8, int16 -- bit-exact
    for shape in [(45, 54, 45), (20, 37, 262), (1, 301, 403), (33, 40, 101)]:
        for dt in (np.float32, np.uint8, np.int16):
            x = (rng.standard_normal(shape) * 40 + 100).astype(dt)
            xd = gpu.asarray(x)
            for mode in ("reflect", "mirror", "nearest", "wrap", "constant"):
                for size in (3, 5, 7, (1, 3, 5) if shape[0] > 1 else (1, 5, 3)):
                    for fn, rf in ((ndi.maximum_filter, sndi.maximum_filter), (ndi.minimum_filter, sndi.minimum_filter)):
                        got = fn(xd, size, mode=mode, cval=7).get()
                        assert np.array_equal(got, rf(x, size, mode=mode, cval=7)), (shape, dt, mode, size, fn.__name__, last_kernel())
            assert np.array_equal(ndi.grey_erosion(xd, size=3).get(), sndi.grey_erosion(x, size=3))
    # integer box filters and the 3 x 3 median on such rows
    for shape in [(301, 403), (20, 37, 262), (45, 54, 45)]:
        for dt in (np.uint8, np.int16, np.float32):
            xi = (rng.standard_normal(shape) * 40 + 100).astype(dt)
            xid = gpu.asarray(xi)
            for mode in ("reflect", "mirror", "nearest", "wrap", "constant"):
                if dt != np.float32:
                    for size in (3, 5, (1,) * (len(shape) - 2) + (3, 7)):
                        assert np.array_equal(ndi.uniform_filter(xid, size, mode=mode, cval=9).get(), sndi.uniform_filter(xi, size, mode=mode, cval=9)), (shape, dt, mode, size)
                fp = np.ones((1,) * (len(shape) - 2) + (3, 3), bool)
                assert np.array_equal(ndi.median_filter(xid, footprint=fp, mode=mode, cval=9).get(), sndi.median_filter(xi, footprint=fp, mode=mode, cval=9)), (shape, dt, mode)
    # a cval the array's dtype does not hold exactly (SciPy uses it as a double): still SciPy's result (generic route)
    xi = (rng.standard_normal((20, 37, 101)) * 40 + 100).astype(np.uint8)
    w3 = rng.standard_normal((3, 3, 3))
    for cv in (3.7, -1.0, 300.0):
        assert np.array_equal(ndi.correlate(gpu.asarray(xi), w3, mode="constant", cval=cv).get(), sndi.correlate(xi, w3, mode="constant", cval=cv)), cv
    xf = rng.standard_normal((20, 37, 101)).astype(np.float32)
    assert np.array_equal(ndi.correlate(gpu.asarray(xf), w3, mode="constant", cval=0.1).get(), sndi.correlate(xf, w3, mode="constant", cval=0.1))
    got = ndi.uniform_filter(gpu.asarray(xf), 5, mode="constant", cval=0.1).get()
    assert np.abs(got - sndi.uniform_filter(xf.astype(np.float64), 5, mode="constant", cval=0.1)).max() <= 1e-6 * 4
    # binary erosion / dilation, one iteration: rows extended by the border value
    for shape in [(45, 54, 45), (20, 37, 262), (301, 403)]:
        b = rng.random(shape) > 0.4
        bd = gpu.asarray(b)
        for st in (sndi.generate_binary_structure(len(shape), 1), sndi.generate_binary_structure(len(shape), len(shape)), np.ones((3,) * (len(shape) - 1) + (5,), bool)):
            for bv in (0, 1):
                for origin in (0, (0,) * (len(shape) - 1) + (1,)):
                    assert np.array_equal(ndi.binary_erosion(bd, st, border_value=bv, origin=origin).get(), sndi.binary_erosion(b, st, border_value=bv, origin=origin)), (shape, bv, origin)
                    assert np.array_equal(ndi.binary_dilation(bd, st, border_value=bv, origin=origin).get(), sndi.binary_dilation(b, st, border_value=bv, origin=origin)), (shape, bv, origin)
    # dense correlate / convolve (LDS-tiled stencil kernel: SciPy's summation order in double, bit-identical)
    for shape in [(45, 54, 45), (20, 37, 262)]:
        for dt in (np.float32, np.uint8, np.int16):
            x = (rng.standard_normal(shape) * 40 + 100).astype(dt)
            xd = gpu.asarray(x)
            for wshape, origin in (((3, 3, 3), 0), ((3, 5, 7), (0, 1, -2)), ((1, 3, 4), (0, 0, 1))):
                w = rng.standard_normal(wshape)
                for mode in ("reflect", "wrap", "constant", "mirror", "nearest"):
                    got = ndi.correlate(xd, w, mode=mode, cval=3, origin=origin).get()
                    assert np.array_equal(got, sndi.correlate(x, w, mode=mode, cval=3, origin=origin)), (shape, dt, wshape, mode, "correlate")
                    got = ndi.convolve(xd, w, mode=mode, cval=3, origin=origin).get()
                    assert np.array_equal(got, sndi.convolve(x, w, mode=mode, cval=3, origin=origin)), (shape, dt, wshape, mode, "convolve")


def test_affine_zstream_sheared_window_all_angles(gpu, ndi):
    """r4b: the staged window of the z-streaming affine kernel is sheared (every staged input row starts at its own first
    needed column), so in-plane rotations by ANY angle -- and shears, anisotropic scalings, flips -- fit LDS twice per CU.
    Bit-identical to the gather kernel for rotations in steps of 7.5 degrees about both decoupled axes, with shears /
    scalings on top, partial tiles, both tile heights; a voxel the window does not hold gathers for itself (covered by
    the matrices whose row spans exceed the plan's sampled estimate by construction: none may differ either)."""
    from cupyimg_amd import _lib, last_kernel
    lib = _lib.load()
    rng = np.random.default_rng(808)
    shape = (40, 150, 200)
    x = rng.standard_normal(shape).astype(np.float32)
    x[3, 4, 5] = np.inf; x[20, 100, 150] = np.nan
    xd = gpu.asarray(x)
    ctr = (np.array(shape) - 1) / 2.0
    took = 0
    cases = []
    for deg in np.arange(0.0, 360.0, 7.5):
        a = np.deg2rad(deg); c, s = np.cos(a), np.sin(a)
        cases.append((np.array([[1.03, 0, 0], [0, c, -s], [0, s, c]]), None))                          # rotation in (y, x): axis 0 streams
        cases.append((np.array([[c, 0, -s], [0, 0.97, 0], [s, 0, c]]), None))                          # rotation in (z, x): axis 1 streams
    for deg, sy, sx, sh in ((20, 1.3, 0.8, 0.2), (-50, 0.7, 1.2, -0.3), (75, 1.1, 1.1, 0.5), (135, 0.9, 1.4, 0.0), (10, 1.9, 0.6, 1.0)):
        a = np.deg2rad(deg); c, s = np.cos(a), np.sin(a)
        R = np.array([[c, -s], [s, c]]) @ np.array([[sy, sh], [0, sx]])
        M = np.eye(3); M[1:, 1:] = R; M[0, 0] = -1.0                                                   # with a flip along the stream axis
        cases.append((M, (44, 131, 190)))
    for M, oshape in cases:
        off = ctr - M @ (ctr if oshape is None else (np.array(oshape) - 1) / 2.0) + np.array([0.25, -1.5, 2.0])
        lib.mi_debug_set_affine_zstream(0); lib.mi_debug_set_interp_c1(5)
        try:
            want = ndi.affine_transform(xd, M, off, output_shape=oshape, order=1, mode="constant", cval=-2.0).get()
        finally:
            lib.mi_debug_set_affine_zstream(1); lib.mi_debug_set_interp_c1(1)
        for ty in (1, 64):
            lib.mi_debug_set_affine_zstream(ty)
            try:
                got = ndi.affine_transform(xd, M, off, output_shape=oshape, order=1, mode="constant", cval=-2.0).get()
                took += "affine3d_zstream_kernel" in last_kernel() or "affine3d_zrect_kernel" in last_kernel()
            finally:
                lib.mi_debug_set_affine_zstream(1)
            assert np.array_equal(got, want, equal_nan=True), (M.tolist(), ty, last_kernel()[:60], int(np.sum(~((got == want) | (np.isnan(got) & np.isnan(want))))))
    assert took >= len(cases), (took, len(cases))          # the streaming kernel took (at least) one tile height of every case


@pytest.mark.parametrize("zfactor", [1, 0])
def test_cubic_affine_zstream_in_plane(gpu, ndi, zfactor):
    """r4b: order-3 affine transforms on float32 coefficients whose matrix leaves axis 0 to itself (`rotate(volume, a,
    axes=(1, 2))` with scipy's DEFAULT order) stream along z (cubic3_zstream_kernel: planes staged in LDS once per tile, a
    ring of five).  Bit-identical to the gather kernel (cubic3_f32_kernel) in every mode -- taps beyond the array through
    the rectangle (reflect / mirror / nearest / constant / wrap), through the gather path of a wave (grid-constant's cval
    taps, grid-wrap's far side, planes that clash in the ring) --, with steps along z of both signs and below one, output
    shapes that differ from the input's, partial tiles, non-finite coefficients; and within float32 accuracy of scipy.
    r5, zfactor = 1 (the default): cubic3_zfactor_kernel evaluates the in-plane part once per INPUT plane and blends four
    values per voxel -- another order of the sums, so it agrees with the gather kernel to float32 rounding (2e-6 of the
    coefficient range) instead of bit for bit, with the SAME set of non-finite voxels; zfactor = 0 is the r4b kernel."""
    import scipy.ndimage as sndi
    from cupyimg_amd import _lib, last_kernel
    lib = _lib.load()
    lib.mi_debug_set_cubic_zfactor(zfactor)
    try:
        _cubic_zstream_in_plane_body(gpu, ndi, lib, "cubic3_zfactor_kernel" if zfactor else "cubic3_zstream_kernel", not zfactor)
    finally:
        lib.mi_debug_set_cubic_zfactor(1)


def _same_cubic(got, want, exact):
    if exact:
        return np.array_equal(got, want, equal_nan=True)
    fin = np.isfinite(want)
    if not np.array_equal(fin, np.isfinite(got)):
        return False
    if not np.array_equal(np.isnan(got), np.isnan(want)):
        return False
    scale = max(1.0, float(np.abs(want[fin]).max())) if fin.any() else 1.0
    return bool(np.abs(got[fin] - want[fin]).max() <= 2e-6 * scale) and np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)])


def _cubic_zstream_in_plane_body(gpu, ndi, lib, KERN, exact):
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(4242)
    took = 0
    for shape, oshape in (((40, 90, 152), None), ((33, 70, 132), (48, 70, 132)), ((20, 64, 64), (20, 100, 200)), ((64, 64, 64), None), ((37, 81, 100), (41, 97, 131))):
        x = rng.standard_normal(shape).astype(np.float32)
        xd = gpu.asarray(x)
        osh = shape if oshape is None else oshape
        for deg, m00, sc in ((7, 1.0, 1.0), (3, 0.9, 1.1), (-10, -1.0, 0.95), (0, 0.5, 0.8), (12, 1.0, 1.0), (90, 1.0, 1.0), (180, 0.7, 1.0)):
            a = np.deg2rad(deg); c, s = np.cos(a), np.sin(a)
            for M in (np.array([[m00, 0, 0], [0, c * sc, -s], [0, s, c * sc]]),              # in the (y, x) plane: axis 0 streams
                      np.array([[c * sc, 0, -s], [0, m00, 0], [s, 0, c * sc]])):             # in the (z, x) plane: axis 1 streams
              off = (np.array(shape) - 1) / 2 - M @ ((np.array(osh) - 1) / 2) + np.array([0.3, -1.7, 2.2])
              for mode in ("constant", "nearest", "mirror", "reflect", "grid-wrap", "grid-constant", "wrap"):
                if M[0, 2] != 0 and mode in ("reflect", "wrap") and deg not in (7, -10):
                    continue
                for prefilter in ((True, False) if mode in ("constant", "mirror") and deg == 7 else (True,)):
                    kw = dict(output_shape=osh, order=3, mode=mode, cval=0.5, prefilter=prefilter)
                    lib.mi_debug_set_cubic_zstream(0); lib.mi_debug_set_cubic_box(0)          # the comparator: the gather kernel
                    try:
                        want = ndi.affine_transform(xd, M, off, **kw).get()
                    finally:
                        lib.mi_debug_set_cubic_zstream(1); lib.mi_debug_set_cubic_box(1)
                    lib.mi_debug_set_cubic_zstream(1 + 4 + 8)          # any angle, and the grid modes (not taken by default)
                    try:
                        got = ndi.affine_transform(xd, M, off, **kw).get()
                        took += KERN in last_kernel()
                    finally:
                        lib.mi_debug_set_cubic_zstream(1)
                    assert _same_cubic(got, want, exact), (shape, osh, deg, m00, mode, prefilter, last_kernel()[:40], int(np.sum(got != want)))
                    if prefilter:
                        ref = sndi.affine_transform(x.astype(np.float64), M, off, output_shape=osh, order=3, mode=mode, cval=0.5)
                        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (shape, deg, mode)
    assert took >= 250, took
    # every wave on the kernel's gather path (the debug value 3): the same bits again, for both stream axes
    x = rng.standard_normal((40, 90, 152)).astype(np.float32); xd = gpu.asarray(x)
    a = np.deg2rad(21.0); M = np.array([[1.0, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    off = np.array([0.0, 20.0, -14.0])
    for Mg, og, name in ((np.array([[np.cos(a), 0, -np.sin(a)], [0, 0.9, 0], [np.sin(a), 0, np.cos(a)]]), np.array([9.0, 2.0, -5.0]), KERN + "<1>"),
                         (M, off, KERN + "<0>")):
        want = ndi.affine_transform(xd, Mg, og, order=3, prefilter=False).get()
        assert name in last_kernel(), last_kernel()
        lib.mi_debug_set_cubic_zstream(3)
        try:
            got = ndi.affine_transform(xd, Mg, og, order=3, prefilter=False).get()
        finally:
            lib.mi_debug_set_cubic_zstream(1)
        assert _same_cubic(got, want, exact)
    # non-finite coefficients stay inside their 4 x 4 x 4 window in both kernels alike
    x[5, 40, 70] = np.inf; x[30, 10, 100] = np.nan
    xd = gpu.asarray(x)
    got = ndi.affine_transform(xd, M, off, order=3, prefilter=False).get()
    lib.mi_debug_set_cubic_zstream(0); lib.mi_debug_set_cubic_box(0)
    try:
        want = ndi.affine_transform(xd, M, off, order=3, prefilter=False).get()
    finally:
        lib.mi_debug_set_cubic_zstream(1); lib.mi_debug_set_cubic_box(1)
    assert _same_cubic(got, want, exact)
    assert np.isfinite(got).sum() > 0.99 * got.size
    # the defaults: a quarter turn (LDS bank conflicts down the columns) and the grid modes are not streamed: they take the box
    # kernel (r5: the LDS-staged box per 16^3 tile takes what the streaming kernels refuse) or the gather kernel
    a = np.deg2rad(90.0); M9 = np.array([[1.0, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    ndi.affine_transform(xd, M9, np.array([0.0, 0.0, 89.0]), order=3, prefilter=False)
    assert "cubic3_f32_kernel" in last_kernel() or "cubic3_box_kernel" in last_kernel()
    ndi.affine_transform(xd, M, off, order=3, prefilter=False, mode="grid-wrap")
    assert "cubic3_f32_kernel" in last_kernel() or "cubic3_box_kernel" in last_kernel()
    ndi.affine_transform(xd, M, off, order=3, prefilter=False, mode="reflect")
    assert KERN in last_kernel()
    # a matrix that couples axis 0, a diagonal one and a float64 array are not taken
    M2 = M.copy(); M2[0, 1] = 0.01
    ndi.affine_transform(xd, M2, off, order=3, prefilter=False)
    assert KERN not in last_kernel()


def test_cubic_affine_rowblend_default_rotate(gpu, ndi):
    """r4b: order-3 affine transforms on float32 coefficients whose matrix leaves the x axis to itself (unit step, integral
    shift) -- `rotate(volume, a)` with SciPy's DEFAULT axes and order -- blend sixteen input rows per output row at a
    uniform row base (cubic3_rowblend_kernel).  Bit-identical to the gather kernel in every mode, for shifts along x of
    both signs that push columns outside, rows longer than a wave's 512 voxels, output shapes that differ from the
    input's; within float32 accuracy of SciPy; `rotate` with the defaults goes through it."""
    # (r5: with prefilter=True the x axis is now left unfiltered and read as one tap -- tests/test_gpu_spline_fast.py::
    # test_axes_that_hold_samples; the bit-identity to the gather kernel below is that of the FULL route: every axis filtered)
    from cupyimg_amd.scipy.ndimage import interpolation as _I
    _I._IDENT_AXES = False
    try:
        _rowblend_full_route_body(gpu, ndi)
    finally:
        _I._IDENT_AXES = True


def _rowblend_full_route_body(gpu, ndi):
    import scipy.ndimage as sndi
    from cupyimg_amd import _lib, last_kernel
    lib = _lib.load()
    rng = np.random.default_rng(777)
    took = 0
    for shape, oshape in (((40, 90, 152), None), ((33, 70, 132), (48, 75, 132)), ((20, 64, 64), (30, 100, 64)), ((30, 50, 1100), None), ((37, 81, 100), (41, 97, 100)),
                          ((45, 55, 181), None), ((31, 44, 70), (40, 50, 67))):          # rows that are not a multiple of four
        x = rng.standard_normal(shape).astype(np.float32)
        xd = gpu.asarray(x)
        osh = shape if oshape is None else oshape
        for deg, sc, xs in ((7, 1.0, 0), (3, 1.1, 2), (-10, 0.95, -3), (45, 0.8, 0), (90, 1.0, 1), (170, 1.0, -60), (12, 1.0, 200)):
            a = np.deg2rad(deg); c, s = np.cos(a), np.sin(a)
            M = np.array([[c * sc, -s, 0], [s, c * sc, 0], [0, 0, 1.0]])
            off = (np.array(shape) - 1) / 2 - M @ ((np.array(osh) - 1) / 2) + np.array([0.3, -1.7, 0.0])
            off[2] = xs
            for mode in ("constant", "nearest", "mirror", "reflect", "grid-wrap", "grid-constant", "wrap"):
                kw = dict(output_shape=osh, order=3, mode=mode, cval=0.5)
                lib.mi_debug_set_cubic_rowblend(0); lib.mi_debug_set_cubic_box(0)          # the comparator: the gather kernel
                try:
                    want = ndi.affine_transform(xd, M, off, **kw).get()
                finally:
                    lib.mi_debug_set_cubic_rowblend(1); lib.mi_debug_set_cubic_box(1)
                got = ndi.affine_transform(xd, M, off, **kw).get()
                took += "cubic3_rowblend_kernel" in last_kernel()
                assert np.array_equal(got, want, equal_nan=True), (shape, osh, deg, xs, mode, last_kernel()[:40], int(np.sum(got != want)))
                if deg in (7, 170):
                    ref = sndi.affine_transform(x.astype(np.float64), M, off, output_shape=osh, order=3, mode=mode, cval=0.5)
                    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (shape, deg, mode)
    assert took == 7 * 7 * 7, took
    v = rng.standard_normal((48, 120, 96)).astype(np.float32)
    v[7, 30, 40] = np.nan
    vd = gpu.asarray(v)
    got = ndi.rotate(vd, 17.0, reshape=False).get()
    assert "cubic3_rowblend_kernel" in last_kernel()
    lib.mi_debug_set_cubic_rowblend(0)
    try:
        want = ndi.rotate(vd, 17.0, reshape=False).get()
    finally:
        lib.mi_debug_set_cubic_rowblend(1)
    assert np.array_equal(got, want, equal_nan=True)
    v[7, 30, 40] = 0.25
    ref = sndi.rotate(v.astype(np.float64), 17.0, reshape=False)
    assert np.abs(ndi.rotate(gpu.asarray(v), 17.0, reshape=False).get() - ref).max() <= 2e-5 * np.abs(ref).max()
    # a fractional shift along x, a non-unit x step and a matrix that mixes x in are not taken
    M = np.array([[np.cos(0.2), -np.sin(0.2), 0], [np.sin(0.2), np.cos(0.2), 0], [0, 0, 1.0]])
    for Mx, offx in ((M, 0.5), (M * np.array([1, 1, 1.01]), 0.0), (M + np.array([[0, 0, 0.01], [0, 0, 0], [0, 0, 0]]), 0.0)):
        ndi.affine_transform(vd, Mx, np.array([1.0, -2.0, offx]), order=3, prefilter=False)
        assert "cubic3_rowblend_kernel" not in last_kernel(), (Mx.tolist(), offx)


def test_affine_rowblend_kernel(gpu, ndi):
    """Matrices that leave the x axis to itself with unit step and an integral shift (a rotation / shear / scaling in the
    (z, y) plane: `rotate(volume, angle)` with the default axes) blend four input ROWS per output row
    (affine3d_rowblend_kernel: no gathers, no LDS).  Bit-identical to the L1-gather kernel; x shifts of both signs that push
    columns outside, partial tiles along x / y / z, exact boundary hits, non-finite samples; a fractional x shift or a
    non-unit x step is not taken; `rotate(order=1)` of a volume goes through it."""
    from cupyimg_amd import _lib, last_kernel
    lib = _lib.load()
    rng = np.random.default_rng(600)

    def zy(deg, s0=1.0, s1=1.0, shear=0.0):
        a = np.deg2rad(deg); c, s = np.cos(a), np.sin(a)
        M = np.eye(3)
        M[0, 0], M[0, 1], M[1, 0], M[1, 1] = c * s0, -s * s1 + shear, s * s0, c * s1
        return M

    cases = [
        ((64, 72, 264), zy(7), np.array([2.0, -3.5, 0.0]), None, True),
        ((64, 72, 264), zy(-30, 1.1, 0.9), np.array([10.0, 20.0, 3.0]), (50, 90, 260), True),
        ((40, 50, 132), zy(45, shear=0.2), np.array([-5.0, 30.0, -7.0]), (70, 66, 140), True),          # x shift pushes 7 columns outside, ox not a multiple of 4 x 64
        ((64, 64, 64), zy(0, shear=1e-9), np.array([1.0, -2.0, 5.0]), None, True),                        # integer shifts: (all but) exact boundary hits
        ((64, 64, 64), zy(180, shear=1e-9), np.array([63.0, 63.0, 0.0]), None, True),                     # flips
        ((64, 64, 64), np.eye(3), np.array([1.0, -2.0, 5.0]), None, "affine3d_z"),                          # diagonal: axis 0 is decoupled too, the z-streaming kernel comes first
        ((64, 72, 264), zy(7), np.array([2.0, -3.5, 0.5]), None, False),                                 # fractional x shift
        ((64, 72, 264), zy(7) @ np.diag([1.0, 1.0, 1.25]), np.array([2.0, -3.5, 0.0]), None, False),     # x step 1.25
    ]
    for shape, M, off, oshape, takes in cases:
        x = rng.standard_normal(shape).astype(np.float32)
        x[10, 9, 11] = np.inf; x[11, 12, 10] = np.nan
        xd = gpu.asarray(x)
        oshape = shape if oshape is None else oshape
        lib.mi_debug_set_interp_c1(5)
        try:
            want = ndi.affine_transform(xd, M, off, output_shape=oshape, order=1, mode="constant", cval=-0.75).get()
        finally:
            lib.mi_debug_set_interp_c1(1)
        got = ndi.affine_transform(xd, M, off, output_shape=oshape, order=1, mode="constant", cval=-0.75).get()
        if isinstance(takes, str):
            assert takes in last_kernel(), (shape, last_kernel())
        else:
            assert ("rowblend" in last_kernel()) == takes, (shape, last_kernel())
        assert np.array_equal(got, want, equal_nan=True), (shape, float(np.nanmax(np.abs(got - want))))
        ref = orc.affine_transform(x, M, off, output_shape=oshape, order=1, mode="constant", cval=-0.75)
        ok = np.isfinite(ref)
        assert np.array_equal(np.isfinite(want), ok), shape
        assert np.allclose(want[ok], ref[ok], rtol=0, atol=2e-6 * max(1.0, np.abs(ref[ok]).max())), shape
    import scipy.ndimage as sndi
    v = rng.standard_normal((72, 80, 260)).astype(np.float32)
    got = ndi.rotate(gpu.asarray(v), 11.0, reshape=False, order=1).get()
    assert "rowblend" in last_kernel(), last_kernel()
    ref = sndi.rotate(v.astype(np.float64), 11.0, reshape=False, order=1)
    assert np.allclose(got, ref, rtol=0, atol=2e-6 * max(1.0, np.abs(ref).max()))


def test_affine_lds_box_kernel_tile_shapes(gpu, ndi):
    """r5: the LDS-staged order-1 affine kernel (matrices that couple all three axes) on every tile shape -- 64 x 8 x 8,
    32 x 16 x 8, the cube 16 x 16 x 16 and 16 x 32 x 8 -- and with boxes up to the 64 KiB budget (rotations up to ~25 degrees
    about a general axis, where the L1-gather kernel has fallen to 0.2 of the roofline): bit-identical to the gather kernel
    (same splits, same blend), volumes that are not multiples of the tile, output shapes that differ, partial boxes at the
    array's faces; the default rule takes the LDS kernel there."""
    from cupyimg_amd import _lib, last_kernel
    lib = _lib.load()
    rng = np.random.default_rng(99)

    def rot(axis, deg):
        a = np.deg2rad(deg); u = np.asarray(axis, float); u /= np.linalg.norm(u)
        K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
        return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)

    for shape, oshape in (((96, 112, 128), None), ((70, 90, 144), (80, 100, 160)), ((64, 64, 256), None)):
        x = rng.standard_normal(shape).astype(np.float32)
        xd = gpu.asarray(x)
        osh = shape if oshape is None else oshape
        for axis, deg in (((1, 1, 1), 12.0), ((1, 1, 0), 22.0), ((1, 0, 1), 5.0), ((0.3, 1, -0.5), 18.0)):
            M = rot(axis, deg) @ np.diag([1.03, 0.97, 1.0])
            off = (np.array(shape) - 1) / 2 - M @ ((np.array(osh) - 1) / 2) + np.array([0.4, -1.3, 2.2])
            kw = dict(output_shape=osh, order=1, mode="constant", cval=0.75)
            lib.mi_debug_set_interp_c1(5)
            try:
                want = ndi.affine_transform(xd, M, off, **kw).get()
                assert "affine3d_c1_kernel" in last_kernel(), last_kernel()
            finally:
                lib.mi_debug_set_interp_c1(1)
            took = 0
            for knob in (0, 1064, 2064, 3064, 4064, 3128):
                lib.mi_debug_set_affine_box_kib(knob)
                try:
                    got = ndi.affine_transform(xd, M, off, **kw).get()
                    took += "affine3d_lds_kernel" in last_kernel()
                finally:
                    lib.mi_debug_set_affine_box_kib(0)
                assert np.array_equal(got, want), (shape, osh, axis, deg, knob, last_kernel()[:70], int(np.sum(got != want)))
            assert took >= 3, (shape, axis, deg, took)
