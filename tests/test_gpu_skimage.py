"""skimage facade on the device: openings / closings / top-hats with the
reference's literal cases (skimage/morphology/grey.py docstrings, tests/test_grey.py:83-125),
eccentric (even-sided) elements against a NumPy/SciPy restatement, and the structural
similarity index against a NumPy restatement of _structural_similarity.py on SciPy's filters."""
import numpy as np
import pytest
import scipy.ndimage as sndi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def skm(gpu):
    from cupyimg_amd.skimage import morphology
    return morphology


@pytest.fixture(scope="module")
def metrics(gpu):
    from cupyimg_amd.skimage import metrics
    return metrics


# ---------------------------------------------------------------- host restatements
def _shift(selem, sx, sy):
    if selem.ndim != 2:
        return selem
    m, n = selem.shape
    if m % 2 == 0:
        z = np.zeros((1, n), selem.dtype)
        selem = np.vstack((selem, z)) if sx else np.vstack((z, selem))
        m += 1
    if n % 2 == 0:
        z = np.zeros((m, 1), selem.dtype)
        selem = np.hstack((selem, z)) if sy else np.hstack((z, selem))
    return selem


def _ero(x, s, sx=False, sy=False):
    return sndi.grey_erosion(x, footprint=_shift(s, sx, sy))


def _dil(x, s, sx=False, sy=False):
    s = _shift(s, sx, sy)
    return sndi.grey_dilation(x, footprint=s[(slice(None, None, -1),) * s.ndim])


def _padded(func):
    def run(x, s):
        w = [n - 1 if n % 2 == 0 else 0 for n in s.shape]
        if not any(w):
            return func(x, s)
        y = func(np.pad(x, [(k, k) for k in w], mode="edge"), s)
        return y[tuple(slice(k, n - k) for k, n in zip(w, y.shape))]
    return run


_open = _padded(lambda x, s: _dil(_ero(x, s), s, True, True))
_close = _padded(lambda x, s: _ero(_dil(x, s), s, True, True))


def test_docstring_cases(gpu, skm):
    bad = np.array([[1, 0, 0, 0, 1], [1, 1, 0, 1, 1], [1, 1, 1, 1, 1], [1, 1, 0, 1, 1], [1, 0, 0, 0, 1]], np.uint8)
    want = np.array([[0, 0, 0, 0, 0], [1, 1, 0, 1, 1], [1, 1, 0, 1, 1], [1, 1, 0, 1, 1], [0, 0, 0, 0, 0]], np.uint8)
    assert np.array_equal(skm.opening(gpu.asarray(bad), skm.square(3)).get(), want)
    broken = np.zeros((5, 5), np.uint8)
    broken[2] = [1, 1, 0, 1, 1]
    want = np.zeros((5, 5), np.uint8)
    want[2] = 1
    assert np.array_equal(skm.closing(gpu.asarray(broken), skm.square(3)).get(), want)
    bright = np.array([[2, 3, 3, 3, 2], [3, 4, 5, 4, 3], [3, 5, 9, 5, 3], [3, 4, 5, 4, 3], [2, 3, 3, 3, 2]], np.uint8)
    want = np.array([[0, 0, 0, 0, 0], [0, 0, 1, 0, 0], [0, 1, 5, 1, 0], [0, 0, 1, 0, 0], [0, 0, 0, 0, 0]], np.uint8)
    assert np.array_equal(skm.white_tophat(gpu.asarray(bright), skm.square(3)).get(), want)
    dark = (11 - bright).astype(np.uint8)
    assert np.array_equal(skm.black_tophat(gpu.asarray(dark), skm.square(3)).get(), want)


def test_pixel_cases(gpu, skm):
    # tests/test_grey.py:71-125 (eccentric structuring elements)
    black = 255 * np.ones((4, 4), np.uint8)
    black[1, 1] = 0
    white = 255 - black
    for s in [skm.square(2), skm.rectangle(2, 2), skm.rectangle(2, 1), skm.rectangle(1, 2)]:
        b, w = gpu.asarray(black), gpu.asarray(white)
        assert np.array_equal(skm.erosion(b, s).get(), 255 - skm.dilation(w, s).get())
        assert np.array_equal(skm.opening(b, s).get(), black)
        assert np.array_equal(skm.closing(w, s).get(), white)
        assert not skm.opening(w, s).get().any()
        assert (skm.closing(b, s).get() == 255).all()
        assert np.array_equal(skm.white_tophat(w, s).get(), white)
        assert np.array_equal(skm.black_tophat(b, s).get(), 255 - black)
        assert not skm.white_tophat(b, s).get().any()
        assert not skm.black_tophat(w, s).get().any()


@pytest.mark.parametrize("dtype", ["uint8", "float32", "int16", "bool"])
def test_open_close_tophat_match_restatement(gpu, skm, dtype):
    rng = np.random.default_rng(140)
    x = rng.random((37, 45)) > 0.5 if dtype == "bool" else (rng.random((37, 45)) * 200).astype(dtype)
    xd = gpu.asarray(x)
    selems = [np.ones((3, 3), np.uint8), np.ones((2, 2), np.uint8), np.ones((4, 3), np.uint8), np.ones((2, 5), np.uint8),
              (rng.random((5, 5)) > 0.4).astype(np.uint8), (rng.random((4, 6)) > 0.3).astype(np.uint8)]
    for s in selems:
        if not s.any():
            continue
        o, c = _open(x, s), _close(x, s)
        assert np.array_equal(skm.opening(xd, s).get(), o), s.shape
        assert np.array_equal(skm.closing(xd, s).get(), c), s.shape
        wt = np.logical_xor(x, o) if dtype == "bool" else (x - o).astype(dtype)
        bt = np.logical_xor(c, x) if dtype == "bool" else (c - x).astype(dtype)
        assert np.array_equal(skm.white_tophat(xd, s).get(), wt)
        assert np.array_equal(skm.black_tophat(xd, s).get(), bt)
    # default element, 3-D, `out`
    v = (rng.random((9, 10, 11)) * 100).astype(np.uint8)
    cross = sndi.generate_binary_structure(3, 1)
    out = gpu.empty(v.shape, np.uint8)
    res = skm.opening(gpu.asarray(v), out=out)
    assert res is out
    assert np.array_equal(out.get(), sndi.grey_dilation(sndi.grey_erosion(v, footprint=cross), footprint=cross))


def test_binary_opening_closing(gpu, skm):
    rng = np.random.default_rng(141)
    x = rng.random((40, 33)) > 0.45
    for s in [np.ones((3, 3), np.uint8), sndi.generate_binary_structure(2, 1).astype(np.uint8), np.ones((5, 3), np.uint8)]:
        eroded = sndi.binary_erosion(x, structure=s, border_value=True)
        assert np.array_equal(skm.binary_opening(gpu.asarray(x), s).get(), sndi.binary_dilation(eroded, structure=s))
        dilated = sndi.binary_dilation(x, structure=s)
        assert np.array_equal(skm.binary_closing(gpu.asarray(x), s).get(),
                              sndi.binary_erosion(dilated, structure=s, border_value=True))
    assert skm.binary_opening(gpu.asarray(x)).get().dtype == np.bool_


def test_selem_generators_on_device(gpu, skm):
    assert np.array_equal(skm.diamond(1).get(), sndi.generate_binary_structure(2, 1))
    assert skm.cube(3).shape == (3, 3, 3) and skm.ball(2).get().sum() == 33
    assert skm.octagon(5, 3).shape == (11, 11) and skm.star(4).shape == (13, 13)
    assert skm.disk(3, dtype=np.bool_).dtype == np.bool_


# ---------------------------------------------------------------- structural similarity
def _ssim_host(x, y, win_size=None, gaussian_weights=False, data_range=None, K1=0.01, K2=0.03, sigma=1.5,
               use_sample_covariance=True, dtype=np.float64, gradient=False):
    truncate = 3.5
    if win_size is None:
        win_size = 2 * int(truncate * sigma + 0.5) + 1 if gaussian_weights else 7
    if data_range is None:
        data_range = 2 if x.dtype.kind == "f" else float(np.iinfo(x.dtype).max) - float(np.iinfo(x.dtype).min)
    if gaussian_weights:
        def filt(a):
            return sndi.gaussian_filter(a, sigma=sigma, truncate=truncate, mode="reflect")
    else:
        def filt(a):
            return sndi.uniform_filter(a, size=win_size, mode="reflect")
    x, y = x.astype(dtype), y.astype(dtype)
    NP = win_size ** x.ndim
    cov = NP / (NP - 1) if use_sample_covariance else 1.0
    ux, uy = filt(x), filt(y)
    uxx, uyy, uxy = filt(x * x), filt(y * y), filt(x * y)
    vx, vy, vxy = cov * (uxx - ux * ux), cov * (uyy - uy * uy), cov * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    A1, A2, B1, B2 = 2 * ux * uy + C1, 2 * vxy + C2, ux ** 2 + uy ** 2 + C1, vx + vy + C2
    D = B1 * B2
    Smap = (A1 * A2) / D
    pad = (win_size - 1) // 2
    inner = Smap[tuple(slice(pad, n - pad) for n in Smap.shape)]
    if not gradient:
        return inner.mean(dtype=np.float64), Smap
    grad = filt(A1 / D) * x + filt(-Smap / B2) * y + filt((ux * (A2 - A1) - uy * (B2 - B1) * Smap) / D)
    return inner.mean(dtype=np.float64), Smap, grad * (2 / x.size)


def test_ssim_matches_restatement(gpu, metrics):
    rng = np.random.default_rng(150)
    for shape in [(64, 72), (20, 24, 28), (40,)]:
        x = (rng.random(shape) * 255).astype(np.uint8)
        y = np.clip(x + rng.normal(0, 20, shape), 0, 255).astype(np.uint8)
        xd, yd = gpu.asarray(x), gpu.asarray(y)
        for kw in [{}, {"win_size": 3}, {"gaussian_weights": True}, {"use_sample_covariance": False, "K1": 0.02},
                   {"data_range": 255, "win_size": 5}]:
            want, wmap = _ssim_host(x, y, **kw)
            got, gmap = metrics.structural_similarity(xd, yd, full=True, **kw)
            assert abs(got - want) <= 1e-10, (shape, kw)
            np.testing.assert_allclose(gmap.get(), wmap, rtol=1e-9, atol=1e-11)
            assert metrics.structural_similarity(xd, yd, **kw) == got
        assert metrics.structural_similarity(xd, xd) == 1.0
    # float32 moments (the fused separable kernel on volumes)
    x = rng.random((48, 40, 56)).astype(np.float32)
    y = (x + 0.1 * rng.standard_normal(x.shape)).astype(np.float32)
    want, wmap = _ssim_host(x, y, data_range=1.0, dtype=np.float32)
    got, gmap = metrics.structural_similarity(gpu.asarray(x), gpu.asarray(y), data_range=1.0, data_dtype=np.float32, full=True)
    assert gmap.dtype == np.float32
    assert abs(got - want) < 2e-5
    np.testing.assert_allclose(gmap.get(), wmap, rtol=0, atol=2e-4)


def test_ssim_gradient_and_multichannel(gpu, metrics):
    rng = np.random.RandomState(5)
    x, y = rng.rand(30, 30) * 255, rng.rand(30, 30) * 255
    want, wmap, wgrad = _ssim_host(x, y, data_range=255, gradient=True)
    got, ggrad, gmap = metrics.structural_similarity(gpu.asarray(x), gpu.asarray(y), data_range=255, gradient=True, full=True)
    assert abs(got - want) < 1e-12 and got < 0.05
    np.testing.assert_allclose(ggrad.get(), wgrad, rtol=1e-8, atol=1e-14)
    np.testing.assert_allclose(gmap.get(), wmap, rtol=1e-9, atol=1e-12)
    g2 = metrics.structural_similarity(gpu.asarray(x), gpu.asarray(y), data_range=255, gradient=True)
    assert g2[0] == got and (g2[1].get() < 0.05).all()
    # channels filtered independently, then averaged
    xc = (rng.rand(32, 36, 3) * 255).astype(np.uint8)
    yc = np.clip(xc + rng.normal(0, 15, xc.shape), 0, 255).astype(np.uint8)
    per = [_ssim_host(xc[..., c], yc[..., c]) for c in range(3)]
    got, gmap = metrics.structural_similarity(gpu.asarray(xc), gpu.asarray(yc), multichannel=True, full=True)
    assert abs(got - np.mean([p[0] for p in per])) < 1e-10
    np.testing.assert_allclose(gmap.get(), np.stack([p[1] for p in per], axis=-1), rtol=1e-9, atol=1e-11)
    with pytest.raises(ValueError):       # win_size exceeds the channel axis when multichannel is off
        metrics.structural_similarity(gpu.asarray(xc), gpu.asarray(yc))


def test_ssim_reference_identities(gpu, metrics):
    """The assertions the reference's own SSIM tests make that need no image file
    (skimage/metrics/tests/test_structural_similarity.py: patch_range :25-33, image :36-58, grad :63-95, dtype :98-112,
    multichannel :115-160, nD :163-172): ssim(X, X) == 1 exactly, decorrelated noise scores low, the full map / the
    gradient have the image's shape, `full` does not change the mean, three identical channels score like one, the
    multichannel mean is the mean of the channels, 1-D ... 4-D inputs work.  The reference draws from CuPy's RandomState;
    the thresholds are statistical, so NumPy's RandomState with the same seeds serves."""
    ssim = metrics.structural_similarity
    rs = np.random.RandomState(1234)
    N = 51
    X = gpu.asarray((rs.rand(N, N) * 255).astype(np.uint8))
    Y = gpu.asarray((rs.rand(N, N) * 255).astype(np.uint8))
    assert ssim(X, Y, win_size=N) < 0.1                       # :31-33
    assert ssim(X, X, win_size=N) == 1
    N = 100
    Xh, Yh = (rs.rand(N, N) * 255).astype(np.uint8), (rs.rand(N, N) * 255).astype(np.uint8)
    X, Y = gpu.asarray(Xh), gpu.asarray(Yh)
    assert ssim(X, X, win_size=3) == 1                        # :42-43
    assert ssim(X, Y, win_size=3) < 0.3
    assert ssim(X, Y, win_size=11, gaussian_weights=True) < 0.3
    mssim0, S3 = ssim(X, Y, full=True)
    assert S3.shape == X.shape                                # :51-54
    assert mssim0 == ssim(X, Y)
    assert ssim(X, X) == 1.0                                  # :57
    for seed in (1, 2, 3, 5, 8, 13):                          # :63-95
        rnd = np.random.RandomState(seed)
        A, B = gpu.asarray(rnd.rand(N, N) * 255), gpu.asarray(rnd.rand(N, N) * 255)
        f = ssim(A, B, data_range=255)
        g = ssim(A, B, data_range=255, gradient=True)
        assert f < 0.05 and g[0] < 0.05 and (g[1].get() < 0.05).all()
        mssim, grad, smap = ssim(A, B, data_range=255, gradient=True, full=True)
        assert (grad.get() < 0.05).all() and grad.shape == A.shape and smap.shape == A.shape
    rs = np.random.RandomState(1234)                          # dtype :98-112
    Xf, Yf = rs.rand(30, 30), rs.rand(30, 30)
    assert ssim(gpu.asarray(Xf), gpu.asarray(Yf)) < 0.15
    X8 = (Xf * 255).astype(np.uint8)
    Y8 = (X8 * 255).astype(np.uint8)                          # (sic: the reference builds Y from X here)
    assert ssim(gpu.asarray(X8), gpu.asarray(Y8)) < 0.15
    S1 = ssim(X, Y, win_size=3)                               # multichannel :115-160
    Xc, Yc = gpu.asarray(np.tile(Xh[..., None], (1, 1, 3))), gpu.asarray(np.tile(Yh[..., None], (1, 1, 3)))
    S2 = ssim(Xc, Yc, multichannel=True, win_size=3)
    assert abs(S1 - S2) < 1e-7
    m, S3 = ssim(Xc, Yc, multichannel=True, full=True)
    assert S3.shape == Xc.shape
    m, grad = ssim(Xc, Yc, multichannel=True, gradient=True)
    assert grad.shape == Xc.shape
    m, grad, S3 = ssim(Xc, Yc, multichannel=True, full=True, gradient=True)
    assert grad.shape == Xc.shape and S3.shape == Xc.shape
    mssim = ssim(Xc, Yc, multichannel=True)
    sep = [float(ssim(gpu.asarray(np.ascontiguousarray(Yc.get()[..., c])), gpu.asarray(np.ascontiguousarray(Xc.get()[..., c]))))
           for c in range(3)]
    assert abs(mssim - np.mean(sep)) < 1e-7
    assert ssim(Xc, Xc, multichannel=True) == 1.0
    with pytest.raises(ValueError):
        ssim(Xc, Yc, win_size=7, multichannel=False)
    rs = np.random.RandomState(7)                             # nD :163-172
    for ndim in range(1, 5):
        # the reference's loop builds [N] * 5 whatever `ndim` is (its xsize does not use the loop variable): 10^5 voxels in 5-D
        A = gpu.asarray((rs.rand(*([10] * 5)) * 255).astype(np.uint8))
        B = gpu.asarray((rs.rand(*([10] * 5)) * 255).astype(np.uint8))
        assert ssim(A, B, win_size=3) < 0.05
        # ... and what it meant to build: 1-D ... 4-D (10 ... 10^4 samples: a loose bound)
        A = gpu.asarray((rs.rand(*([10] * ndim)) * 255).astype(np.uint8))
        B = gpu.asarray((rs.rand(*([10] * ndim)) * 255).astype(np.uint8))
        assert ssim(A, B, win_size=3) < 0.6, ndim


def test_ssim_errors_and_warnings(gpu, metrics):
    X = gpu.zeros((9, 9), np.float64)
    with pytest.raises(ValueError):
        metrics.structural_similarity(X, gpu.zeros((8, 8), np.float64))
    with pytest.raises(ValueError):
        metrics.structural_similarity(X, X, win_size=10)
    with pytest.raises(ValueError):
        metrics.structural_similarity(X, X, win_size=4)
    for bad in ({"K1": -0.1}, {"K2": -0.1}, {"sigma": -1.0}):
        with pytest.raises(ValueError):
            metrics.structural_similarity(X, X, **bad)
    rng = np.random.default_rng(151)
    a = (rng.random((20, 20)) * 255).astype(np.uint8)
    b = (rng.random((20, 20)) * 255).astype(np.uint8)
    base = metrics.structural_similarity(gpu.asarray(a), gpu.asarray(b))
    with pytest.warns(UserWarning, match="mismatched dtype"):
        mixed = metrics.structural_similarity(gpu.asarray(a), gpu.asarray(b.astype(np.float32)))
    assert abs(mixed - base) < 1e-12


def test_simple_metrics(gpu, metrics):
    rng = np.random.default_rng(152)
    a = (rng.random((50, 60)) * 255).astype(np.uint8)
    b = np.clip(a + rng.normal(0, 10, a.shape), 0, 255).astype(np.uint8)
    af, bf = a.astype(np.float64), b.astype(np.float64)
    mse = np.mean((af - bf) ** 2)
    ad, bd = gpu.asarray(a), gpu.asarray(b)
    assert abs(metrics.mean_squared_error(ad, bd) - mse) < 1e-10
    assert abs(metrics.peak_signal_noise_ratio(ad, bd) - 10 * np.log10(255 ** 2 / mse)) < 1e-10
    assert abs(metrics.normalized_root_mse(ad, bd) - np.sqrt(mse) / np.sqrt(np.mean(af * af))) < 1e-12
    assert abs(metrics.normalized_root_mse(ad, bd, normalization="min-max") - np.sqrt(mse) / (af.max() - af.min())) < 1e-12
    assert abs(metrics.normalized_root_mse(ad, bd, normalization="mean") - np.sqrt(mse) / af.mean()) < 1e-12
    with pytest.raises(ValueError):
        metrics.normalized_root_mse(ad, bd, normalization="foo")
    with pytest.raises(ValueError):
        metrics.mean_squared_error(ad, gpu.zeros((3, 3), np.uint8))
    f = rng.random((16, 16)).astype(np.float32) * 3       # outside [-1, 1]
    with pytest.raises(ValueError):
        metrics.peak_signal_noise_ratio(gpu.asarray(f), gpu.asarray(f))


# ------------------------------------------------------------------ r3: committed known-answer fixture
def test_skimage_known_answer_fixture(gpu, skm):
    """tests/golden/skimage_kat.json: the reference's literal vectors (test_grey.py:222-333, test_binary.py:52-58,
    134-148, the grey.py docstrings) replayed through the facade -- float images with assert_allclose's 1e-7, integer
    and bool images exactly, strided `out`, default elements."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "skimage_kat.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 20
    for c in cases:
        img = np.array(c["image"], dtype=c["dtype"])
        want = np.array(c["expected"], dtype=c["expected_dtype"])
        fn = getattr(skm, c["func"])
        args = [gpu.asarray(img)]
        if "selem_ones" in c:
            args.append(gpu.asarray(np.ones(c["selem_ones"], np.uint8)))
        if "out_step" in c:
            big = gpu.zeros(tuple(c["out_big_shape"]), want.dtype)
            view = big[::c["out_step"], ::c["out_step"]]
            fn(*args, out=view)
            got = big.get()
        else:
            got = fn(*args).get()
        assert got.shape == want.shape, c["name"]
        if img.dtype.kind == "f":
            assert np.allclose(got, want, rtol=1e-7, atol=0), c["name"]
        else:
            assert np.array_equal(got.astype(want.dtype), want), c["name"]
            if c["func"].startswith("binary"):
                assert got.dtype == np.bool_, c["name"]
