"""2-D images (the shapes skimage callers pass) through the single-launch streaming kernels: separable float32
filters with 3..17 taps, uint8 / float32 flat min / max, 3 x 3 median -- against scipy.ndimage on the same inputs.
Integer and selection results bit-exact, float32 filters within 1e-6 (max-norm relative)."""
import numpy as np
import pytest
import scipy.ndimage as sndi

from _cases import maxnorm_rel

pytestmark = pytest.mark.gpu
MODES = ["reflect", "constant", "nearest", "mirror", "wrap"]


@pytest.fixture(scope="module")
def ndi(gpu):
    from cupyimg_amd.scipy import ndimage
    return ndimage


@pytest.mark.parametrize("shape", [(37, 64), (150, 264), (61, 512), (300, 1032), (8, 8)])
def test_separable_filters_on_images(gpu, ndi, shape):
    rng = np.random.default_rng(70)
    x = rng.standard_normal(shape).astype(np.float32)
    xd = gpu.asarray(x)
    for mode in MODES:
        for size in (3, 5, 7, 9):
            ref = sndi.uniform_filter(x.astype(np.float64), size, mode=mode, cval=0.75)
            got = ndi.uniform_filter(xd, size, mode=mode, cval=0.75).get()
            assert got.dtype == np.float32
            assert maxnorm_rel(got, ref) <= 1e-6, (shape, "uniform", size, mode)
        for sigma in (0.6, 1.0, 2.0):
            ref = sndi.gaussian_filter(x.astype(np.float64), sigma, mode=mode, cval=-0.5)
            got = ndi.gaussian_filter(xd, sigma, mode=mode, cval=-0.5).get()
            assert maxnorm_rel(got, ref) <= 1e-6, (shape, "gaussian", sigma, mode)
        ref = sndi.sobel(x.astype(np.float64), axis=0, mode=mode, cval=0.25)
        got = ndi.sobel(xd, axis=0, mode=mode, cval=0.25).get()
        assert maxnorm_rel(got, ref) <= 1e-6, (shape, "sobel", mode)


def test_image_kernel_matches_volume_kernel(gpu, ndi):
    """The image path (streaming launch) against the tiled volume kernel it replaced for <= 9 taps."""
    from cupyimg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(71)
    x = rng.standard_normal((200, 520)).astype(np.float32)
    xd = gpu.asarray(x)
    for size, mode in [(3, "reflect"), (5, "mirror"), (7, "wrap"), (9, "nearest"), (5, "constant")]:
        new = ndi.uniform_filter(xd, size, mode=mode, cval=2.0).get()
        lib.mi_debug_set_sep3d_image2d(0)
        try:
            old = ndi.uniform_filter(xd, size, mode=mode, cval=2.0).get()
        finally:
            lib.mi_debug_set_sep3d_image2d(1)
        assert np.abs(new - old).max() <= 1e-6 * np.abs(old).max(), (size, mode)


@pytest.mark.parametrize("shape", [(50, 64), (33, 1040), (120, 2048 + 32), (7, 32)])
def test_uint8_minmax_on_images(gpu, ndi, shape):
    rng = np.random.default_rng(72)
    x = rng.integers(0, 256, size=shape, dtype=np.uint8)
    xd = gpu.asarray(x)
    for mode in MODES:
        for size in (3, 5, 7, 9, (3, 7), (5, 1), (1, 9)):
            for name in ("grey_erosion", "grey_dilation"):
                ref = getattr(sndi, name)(x, size=size, mode=mode, cval=7)
                got = getattr(ndi, name)(xd, size=size, mode=mode, cval=7).get()
                assert got.dtype == np.uint8
                assert np.array_equal(got, ref), (shape, name, size, mode)
    # footprint of ones = size; output argument; maximum_filter
    out = gpu.empty(shape, np.uint8)
    ndi.maximum_filter(xd, footprint=np.ones((5, 5), bool), output=out)
    assert np.array_equal(out.get(), sndi.maximum_filter(x, size=5))


@pytest.mark.parametrize("dtype", ["float32", "uint8", "float64"])
@pytest.mark.parametrize("shape", [(40, 64), (65, 264), (19, 1040), (130, 2048 + 48), (3, 32), (1, 64), (2, 48)])
def test_median3x3(gpu, ndi, dtype, shape):
    rng = np.random.default_rng(73)
    if dtype == "uint8":
        x = rng.integers(0, 256, size=shape, dtype=np.uint8)
        x[::3, ::5] = 255
        x[1::4, 2::7] = 0
    else:
        x = rng.standard_normal(shape).astype(dtype)
        x[::3, ::5] = np.inf
        x[1::4, 2::7] = -np.inf
        x[2::5, 1::3] = x[0, 0]            # ties
    xd = gpu.asarray(x)
    for mode in MODES:
        ref = sndi.median_filter(x, size=3, mode=mode, cval=3)
        got = ndi.median_filter(xd, size=3, mode=mode, cval=3).get()
        assert got.dtype == x.dtype
        assert np.array_equal(got, ref), (dtype, shape, mode)
    assert np.array_equal(ndi.rank_filter(xd, 4, size=3).get(), sndi.rank_filter(x, 4, size=3))
    assert np.array_equal(ndi.percentile_filter(xd, 50, footprint=np.ones((3, 3))).get(), sndi.percentile_filter(x, 50, size=3))
    # other ranks / footprints / origins keep the generic kernel
    assert np.array_equal(ndi.rank_filter(xd, 3, size=3).get(), sndi.rank_filter(x, 3, size=3))
    if shape[0] >= 3:
        assert np.array_equal(ndi.median_filter(xd, size=3, origin=(0, 1)).get(), sndi.median_filter(x, size=3, origin=(0, 1)))


def test_median3x3_planes_of_a_volume(gpu, ndi):
    rng = np.random.default_rng(74)
    x = rng.standard_normal((5, 33, 72)).astype(np.float32)
    u = rng.integers(0, 256, size=(4, 21, 96), dtype=np.uint8)
    for a in (x, u):
        for mode in ("reflect", "constant", "wrap"):
            ref = sndi.median_filter(a, size=(1, 3, 3), mode=mode, cval=1)
            got = ndi.median_filter(gpu.asarray(a), size=(1, 3, 3), mode=mode, cval=1).get()
            assert np.array_equal(got, ref), (a.dtype, mode)


def test_slicewise_and_anisotropic_volumes(gpu, ndi):
    """Volumes filtered slice by slice (no z taps: the image kernel over all planes) and with a z kernel that differs
    from the in-plane one (z pass + fused y/x pass)."""
    rng = np.random.default_rng(75)
    x = rng.standard_normal((21, 45, 264)).astype(np.float32)
    xd = gpu.asarray(x)
    x64 = x.astype(np.float64)
    for mode in MODES:
        for size in [(1, 3, 3), (1, 5, 5), (1, 9, 9), (1, 7, 7)]:
            ref = sndi.uniform_filter(x64, size, mode=mode, cval=1.5)
            got = ndi.uniform_filter(xd, size, mode=mode, cval=1.5).get()
            assert maxnorm_rel(got, ref) <= 1e-6, ("uniform", size, mode)
        for sigma in [(0, 1, 1), (0, 2, 2), (1, 2, 2), (2, 1, 1), (0.5, 2.5, 2.5), (3, 1.5, 1.5)]:
            ref = sndi.gaussian_filter(x64, sigma, mode=mode, cval=-1.0)
            got = ndi.gaussian_filter(xd, sigma, mode=mode, cval=-1.0).get()
            assert maxnorm_rel(got, ref) <= 1e-6, ("gaussian", sigma, mode)
    # per-axis modes and a y origin
    ref = sndi.uniform_filter(x64, (1, 5, 5), mode=["nearest", "wrap", "mirror"], origin=(0, 1, 0))
    got = ndi.uniform_filter(xd, (1, 5, 5), mode=["nearest", "wrap", "mirror"], origin=(0, 1, 0)).get()
    assert maxnorm_rel(got, ref) <= 1e-6


@pytest.mark.parametrize("shape", [(37, 64), (150, 260), (61, 512), (9, 4), (21, 45, 136)])
def test_float64_streaming_kernels(gpu, ndi, shape):
    """float64 images / volumes (skimage's working dtype): separable filters and flat min / max."""
    rng = np.random.default_rng(76)
    x = rng.standard_normal(shape)
    xd = gpu.asarray(x)
    nd = len(shape)
    for mode in MODES:
        for size in (3, 5, 9) if shape[-1] >= 8 else (3,):
            ref = sndi.uniform_filter(x, size, mode=mode, cval=0.75)
            got = ndi.uniform_filter(xd, size, mode=mode, cval=0.75).get()
            assert got.dtype == np.float64
            assert maxnorm_rel(got, ref) <= 1e-12, (shape, "uniform", size, mode)
            for name in ("minimum_filter", "maximum_filter"):
                ref = getattr(sndi, name)(x, size=size, mode=mode, cval=0.25)
                got = getattr(ndi, name)(xd, size=size, mode=mode, cval=0.25).get()
                assert np.array_equal(got, ref), (shape, name, size, mode)
        for sigma in (0.6, 1.0, 2.0, 4.0) if shape[-1] >= 40 else (0.6,):
            ref = sndi.gaussian_filter(x, sigma, mode=mode, cval=-0.5)
            got = ndi.gaussian_filter(xd, sigma, mode=mode, cval=-0.5).get()
            assert maxnorm_rel(got, ref) <= 1e-12, (shape, "gaussian", sigma, mode)
        ref = sndi.sobel(x, axis=nd - 2, mode=mode, cval=0.25)
        got = ndi.sobel(xd, axis=nd - 2, mode=mode, cval=0.25).get()
        assert maxnorm_rel(got, ref) <= 1e-12, (shape, "sobel", mode)
    if nd == 3:
        for sigma in [(0, 1, 1), (1, 2, 2), (2, 1, 0), (1.5, 0, 0)]:
            ref = sndi.gaussian_filter(x, sigma, mode="mirror")
            got = ndi.gaussian_filter(xd, sigma, mode="mirror").get()
            assert maxnorm_rel(got, ref) <= 1e-12, (shape, sigma)
        ref = sndi.grey_erosion(x, size=(3, 5, 5))
        assert np.array_equal(ndi.grey_erosion(xd, size=(3, 5, 5)).get(), ref)
    # origins along y, per-axis modes
    org = (0,) * (nd - 2) + (1, 0)
    modes = ["nearest", "wrap", "mirror"][-nd:]
    ref = sndi.uniform_filter(x, 3, mode=modes, origin=org)
    assert maxnorm_rel(ndi.uniform_filter(xd, 3, mode=modes, origin=org).get(), ref) <= 1e-12


@pytest.mark.parametrize("dtype", ["uint16", "int16"])
@pytest.mark.parametrize("shape", [(50, 64), (33, 520), (33, 528), (40, 1024 + 24), (7, 16), (9, 20, 72)])
def test_16bit_minmax_and_median(gpu, ndi, dtype, shape):
    """uint16 / int16 images and volumes: flat min / max and the 3 x 3 median, bit-exact."""
    rng = np.random.default_rng(77)
    info = np.iinfo(dtype)
    x = rng.integers(info.min, info.max + 1, size=shape, dtype=dtype)
    x[..., ::3, ::5] = info.max
    x[..., 1::4, 2::7] = info.min
    xd = gpu.asarray(x)
    nd = len(shape)
    sizes = [3, 5, 9, (3, 7), (5, 1), (1, 9)] if nd == 2 else [3, 5, (1, 5, 5), (3, 5, 5), (7, 1, 7), (1, 1, 3), (3, 1, 1)]
    for mode in MODES:
        for size in sizes:
            for name in ("grey_erosion", "grey_dilation"):
                ref = getattr(sndi, name)(x, size=size, mode=mode, cval=-7 if dtype == "int16" else 40000)
                got = getattr(ndi, name)(xd, size=size, mode=mode, cval=-7 if dtype == "int16" else 40000).get()
                assert got.dtype == x.dtype
                assert np.array_equal(got, ref), (dtype, shape, name, size, mode)
        msize = 3 if nd == 2 else (1, 3, 3)
        ref = sndi.median_filter(x, size=msize, mode=mode, cval=3)
        got = ndi.median_filter(xd, size=msize, mode=mode, cval=3).get()
        assert np.array_equal(got, ref), (dtype, shape, "median", mode)


def test_blocked_spline_prefilter(gpu, ndi):
    """Orders 2 / 3 on arrays with few long lines: chunks of a line filtered in parallel with a 40-sample horizon.
    Same coefficients as the sequential kernel (and as SciPy) to 1e-13; short chunks forced by the hook."""
    from cupyimg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(78)
    try:
        for shape in [(300, 520), (97, 1031), (600,), (12, 200, 150)]:
            x = rng.standard_normal(shape) * 100.0
            xd = gpu.asarray(x)
            for order in (2, 3):
                for mode in ("mirror", "reflect", "nearest", "constant", "grid-wrap"):
                    ref = sndi.spline_filter(x, order=order, mode=mode)
                    for hook in (0, 16, 50):
                        lib.mi_debug_set_spline_chunk(hook)
                        got = ndi.spline_filter(xd, order=order, mode=mode).get()
                        assert maxnorm_rel(got, ref) <= 1e-13, (shape, order, mode, hook)
            lib.mi_debug_set_spline_chunk(16)
            ref = sndi.spline_filter1d(x, order=3, axis=0, mode="mirror")
            assert maxnorm_rel(ndi.spline_filter1d(xd, order=3, axis=0, mode="mirror").get(), ref) <= 1e-13
            # float32 coefficients (the float32 cubic interpolation route)
            x32 = x.astype(np.float32)
            ref = sndi.shift(x32.astype(np.float64), 0.3, order=3, mode="mirror")
            got = ndi.shift(gpu.asarray(x32), 0.3, order=3, mode="mirror").get()
            assert got.dtype == np.float32 and maxnorm_rel(got, ref) <= 2e-6, shape
    finally:
        lib.mi_debug_set_spline_chunk(0)


def _disk(r):
    y, x = np.mgrid[-r:r + 1, -r:r + 1]
    return (x * x + y * y) <= r * r


@pytest.mark.parametrize("dtype", ["uint8", "uint16", "int16", "float32"])
@pytest.mark.parametrize("shape", [(50, 64), (33, 1040), (90, 2048 + 32), (5, 32), (6, 40, 96)])
def test_uint8_footprints_of_centred_runs(gpu, ndi, shape, dtype):
    """skimage-style footprints (disk, diamond, square, ...) on uint8 / 16-bit images: one streaming launch, bit-exact."""
    rng = np.random.default_rng(79)
    if dtype == "float32":
        x = rng.standard_normal(shape).astype(np.float32)
        x[..., ::5, ::7] = np.inf
        x[..., 2::6, 1::5] = -np.inf
    else:
        info = np.iinfo(dtype)
        x = rng.integers(info.min, info.max + 1, size=shape, dtype=dtype)
    xd = gpu.asarray(x)
    diamond2 = np.abs(np.mgrid[-2:3, -2:3]).sum(0) <= 2
    fps = [_disk(1), _disk(2), _disk(3), _disk(4), diamond2, np.ones((3, 5), bool), np.ones((7, 1), bool),
           np.array([[0, 1, 0], [1, 1, 1], [1, 1, 1]], bool),                       # rows differ top / bottom
           np.array([[1, 1, 1], [0, 0, 0], [1, 1, 1]], bool),                       # an empty row
           np.array([[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [0, 1, 1, 1, 0]], bool),
           np.array([[1, 0, 1], [1, 1, 1], [0, 1, 0]], bool),                       # not runs: generic kernel
           np.array([[1, 1, 0], [1, 1, 1], [0, 1, 0]], bool)]                       # not centred: generic kernel
    for fp in fps:
        f = fp if x.ndim == 2 else fp[None]
        for mode in MODES:
            for name in ("grey_erosion", "grey_dilation", "minimum_filter", "maximum_filter"):
                ref = getattr(sndi, name)(x, footprint=f, mode=mode, cval=9)
                got = getattr(ndi, name)(xd, footprint=f, mode=mode, cval=9).get()
                assert np.array_equal(got, ref), (shape, fp.astype(int).tolist(), name, mode)
    out = gpu.empty(shape, x.dtype)
    f = _disk(2) if x.ndim == 2 else _disk(2)[None]
    ndi.grey_opening(xd, footprint=f, output=out)
    assert np.array_equal(out.get(), sndi.grey_opening(x, footprint=f))


def test_structuring_element_host_hint(gpu, ndi):
    """The footprint constructors remember their host source (no device round trip per morphology call); writing the
    device array drops the hint."""
    from cupyimg_amd import core
    from cupyimg_amd.skimage import morphology as skm
    rng = np.random.default_rng(80)
    x = rng.integers(0, 256, size=(40, 64), dtype=np.uint8)
    xd = gpu.asarray(x)
    d = skm.disk(2)
    assert d._hc is not None and np.array_equal(core.host_copy(d), d.get())
    assert np.array_equal(skm.erosion(xd, d).get(), sndi.grey_erosion(x, footprint=d.get()))
    d[0, 0] = 1                      # now a different element: the hint must not be used any more
    assert d._hc is None
    fp = d.get()
    assert fp[0, 0] == 1
    assert np.array_equal(skm.erosion(xd, d).get(), sndi.grey_erosion(x, footprint=fp))
    v = d[::-1]                      # views never carry the hint
    assert v._hc is None


def test_host_hint_dropped_by_every_write_path(gpu, ndi):
    """Writes through views, `set`, copies into the array and use as `output=` all invalidate the remembered host
    copy of a structuring element (a stale hint made morphology silently use the old mask)."""
    from cupyimg_amd import core
    from cupyimg_amd.skimage import morphology as skm
    writes = [
        lambda d: d[1].__setitem__(2, 0),                       # write through an integer-indexed view
        lambda d: d[1:].__setitem__(Ellipsis, 0),               # ... a sliced view
        lambda d: d.reshape(-1).__setitem__(3, 0),              # ... a reshaped view
        lambda d: d.T.__setitem__(Ellipsis, gpu.asarray(np.zeros(d.shape[::-1], d.dtype))),
        lambda d: d.set(np.zeros(d.shape, d.dtype)),
        lambda d: d[2:3].fill(0),
        lambda d: ndi.grey_erosion(gpu.asarray(np.ones(d.shape, d.dtype)), size=3, output=d),
    ]
    for w in writes:
        d = skm.disk(2)
        assert d._hc is not None
        w(d)
        assert d._hc is None
        assert np.array_equal(core.host_copy(d), d.get())


def test_image_view_cache_is_not_a_reference_cycle(gpu, ndi):
    """ndarray._as3() caches the one-plane-volume view; the device buffer must be released by reference counting
    (the cached view used to point back at the array: freed only when the cyclic collector ran)."""
    import gc
    import weakref
    gc.collect()
    gc.disable()
    try:
        a = gpu.asarray(np.random.default_rng(1).standard_normal((64, 64)).astype(np.float32))
        ndi.uniform_filter(a, 3)                                  # takes the image path -> a._as3()
        assert a._v3 is not None
        mem = weakref.ref(a._mem)
        del a
        assert mem() is None
    finally:
        gc.enable()


def test_median_fast_path_needs_the_full_inplane_window(gpu, ndi):
    """Footprints with nine ones and trailing shape (3, 3) that are NOT the in-plane 3 x 3 window must not take the
    streaming 3 x 3 median (which never sees the footprint)."""
    rng = np.random.default_rng(82)
    x = rng.standard_normal((9, 12, 16)).astype(np.float32)
    xd = gpu.asarray(x)
    fps = []
    f = np.zeros((3, 3, 3), bool); f[:, 1, :] = True; fps.append(f)          # a z-x plane
    f = np.zeros((3, 3, 3), bool); f[:, :, 1] = True; fps.append(f)          # a z-y plane
    f = np.zeros((5, 3, 3), bool); f.reshape(-1)[[0, 5, 9, 13, 20, 22, 31, 40, 44]] = True; fps.append(f)
    f = np.zeros((1, 3, 3), bool); f[...] = True; fps.append(f)              # the one that qualifies
    for f in fps:
        assert f.sum() == 9
        for fn, sfn, kw in [(ndi.median_filter, sndi.median_filter, {}),
                            (ndi.rank_filter, sndi.rank_filter, {"rank": 4}),
                            (ndi.percentile_filter, sndi.percentile_filter, {"percentile": 50})]:
            got = fn(xd, footprint=f, **kw).get()
            assert np.array_equal(got, sfn(x, footprint=f, **kw)), (f.shape, fn.__name__)


@pytest.mark.parametrize("shape", [(20, 30, 64), (9, 17, 1040), (33, 40, 2048 + 32), (3, 3, 32), (1, 5, 48)])
def test_volume_footprints_of_centred_runs(gpu, ndi, shape):
    """6- / 18- / 26-connected structures (and other 3 x 3 x 3 footprints of centred runs) on uint8 volumes: grey
    erosion / dilation in one streaming launch; the same kernel behind binary erosion / dilation of bool volumes."""
    rng = np.random.default_rng(81)
    x = rng.integers(0, 256, size=shape, dtype=np.uint8)
    xd = gpu.asarray(x)
    odd = np.zeros((3, 3, 3), bool)
    odd[0, 1, 1] = odd[1, 0, :] = odd[1, 1, 1] = odd[2, 2, :] = odd[2, 1, 1] = True
    fps = [sndi.generate_binary_structure(3, 1), sndi.generate_binary_structure(3, 2), sndi.generate_binary_structure(3, 3), odd,
           np.ones((3, 3, 1), bool)]
    for fp in fps:
        for mode in MODES:
            for name in ("grey_erosion", "grey_dilation"):
                ref = getattr(sndi, name)(x, footprint=fp, mode=mode, cval=9)
                got = getattr(ndi, name)(xd, footprint=fp, mode=mode, cval=9).get()
                assert np.array_equal(got, ref), (shape, fp.astype(int).tolist(), name, mode)
    b = rng.random(shape) > 0.35
    bd = gpu.asarray(b)
    for st in (None, sndi.generate_binary_structure(3, 2), np.ones((3, 3, 3), bool), odd):
        for bv in (0, 1):
            for it in (1, 2, 3, -1):
                for name in ("binary_erosion", "binary_dilation"):
                    ref = getattr(sndi, name)(b, structure=st, iterations=it, border_value=bv)
                    got = getattr(ndi, name)(bd, structure=st, iterations=it, border_value=bv).get()
                    assert got.dtype == np.bool_ and np.array_equal(got, ref), (shape, name, it, bv)
        assert np.array_equal(ndi.binary_opening(bd, structure=st).get(), sndi.binary_opening(b, structure=st))
        assert np.array_equal(ndi.binary_closing(bd, structure=st, iterations=2).get(), sndi.binary_closing(b, structure=st, iterations=2))


@pytest.mark.parametrize("dtype", ["uint8", "uint16", "int16"])
@pytest.mark.parametrize("shape", [(50, 64), (33, 1040), (90, 2048 + 32), (5, 32), (6, 40, 96)])
def test_uint8_uniform_filter_integer_kernel(gpu, ndi, shape, dtype):
    """uniform_filter on uint8 / 16-bit images with a result of the same dtype: one integer-arithmetic launch, bit-exact
    (SciPy truncates to the integer dtype after every axis)."""
    rng = np.random.default_rng(82)
    info = np.iinfo(dtype)
    x = rng.integers(info.min, info.max + 1, size=shape, dtype=dtype)
    x[..., ::7, ::3] = info.max
    x[..., 1::5, 1::4] = info.min
    xd = gpu.asarray(x)
    lead = (1,) * (x.ndim - 2)
    for mode in MODES:
        for size in [(3, 3), (5, 5), (9, 9), (7, 3), (1, 5), (9, 1), (3, 7)]:
            ref = sndi.uniform_filter(x, lead + size, mode=mode, cval=7)
            got = ndi.uniform_filter(xd, lead + size, mode=mode, cval=7).get()
            assert got.dtype == x.dtype
            assert np.array_equal(got, ref), (dtype, shape, size, mode)
    org = (0,) * (x.ndim - 2) + (1, 0)
    ref = sndi.uniform_filter(x, lead + (5, 3), mode=["nearest", "mirror", "wrap"][-x.ndim:], origin=org)
    got = ndi.uniform_filter(xd, lead + (5, 3), mode=["nearest", "mirror", "wrap"][-x.ndim:], origin=org).get()
    assert np.array_equal(got, ref)
    # worst cases for the division: constant images at every level and sizes
    for val in (info.min, info.min + 1, -1 if info.min < 0 else 1, info.max - 1, info.max):
        c = np.full(shape, val, x.dtype)
        for size in (3, 5, 7, 9):
            assert np.array_equal(ndi.uniform_filter(gpu.asarray(c), lead + (size, size)).get(), sndi.uniform_filter(c, lead + (size, size)))


@pytest.mark.parametrize("dtype", ["float32", "uint8", "int16"])
def test_medians_of_25_and_27_samples(gpu, ndi, dtype):
    """5 x 5 and 3 x 3 x 3 medians: the sorting network with the tap count and the rank fixed at compile time."""
    rng = np.random.default_rng(83)
    if dtype == "float32":
        img = rng.standard_normal((45, 70)).astype(dtype)
        vol = rng.standard_normal((9, 20, 33)).astype(dtype)
        img[::4, ::5] = np.inf
        vol[::2, ::3, ::4] = -np.inf
    else:
        info = np.iinfo(dtype)
        img = rng.integers(info.min, info.max + 1, size=(45, 70), dtype=dtype)
        vol = rng.integers(info.min, info.max + 1, size=(9, 20, 33), dtype=dtype)
    for mode in MODES:
        assert np.array_equal(ndi.median_filter(gpu.asarray(img), size=5, mode=mode, cval=2).get(), sndi.median_filter(img, size=5, mode=mode, cval=2))
        assert np.array_equal(ndi.median_filter(gpu.asarray(vol), size=3, mode=mode, cval=2).get(), sndi.median_filter(vol, size=3, mode=mode, cval=2))
    # the same window with another rank keeps the run-time network
    assert np.array_equal(ndi.rank_filter(gpu.asarray(img), 7, size=5).get(), sndi.rank_filter(img, 7, size=5))
    assert np.array_equal(ndi.percentile_filter(gpu.asarray(vol), 50, size=3).get(), sndi.percentile_filter(vol, 50, size=3))


@pytest.mark.parametrize("dtype", ["uint8", "uint16", "int16"])
def test_integer_uniform_filter_on_volumes(gpu, ndi, dtype):
    """uniform_filter on integer volumes: z pass + fused y/x pass in integer arithmetic, bit-exact."""
    rng = np.random.default_rng(84)
    info = np.iinfo(dtype)
    for shape in [(20, 30, 64), (9, 17, 1040), (5, 6, 32)]:
        x = rng.integers(info.min, info.max + 1, size=shape, dtype=dtype)
        x[::3, ::4, ::5] = info.max
        x[1::3, 1::4, 2::5] = info.min
        xd = gpu.asarray(x)
        for mode in MODES:
            for size in [3, 5, (3, 5, 7), (9, 1, 1), (5, 1, 3), (3, 3, 1), (7, 9, 5)]:
                ref = sndi.uniform_filter(x, size, mode=mode, cval=5)
                got = ndi.uniform_filter(xd, size, mode=mode, cval=5).get()
                assert got.dtype == x.dtype and np.array_equal(got, ref), (dtype, shape, size, mode)
        ref = sndi.uniform_filter(x, (3, 5, 3), mode=["wrap", "nearest", "mirror"], origin=(1, -1, 0))
        assert np.array_equal(ndi.uniform_filter(xd, (3, 5, 3), mode=["wrap", "nearest", "mirror"], origin=(1, -1, 0)).get(), ref)
