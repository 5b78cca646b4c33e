"""scipy.signal direct-convolution callers of the correlate kernel
against scipy.signal itself, plus convolve_separable and the device pad they lean on."""
import numpy as np
import pytest
import scipy.ndimage as sndi
import scipy.signal as ssig

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sig(gpu):
    from cupyimg_amd.scipy import signal
    return signal


def _rand(rng, shape, dtype):
    if np.dtype(dtype).kind == "f":
        return rng.standard_normal(shape).astype(dtype)
    return rng.integers(-9, 10, size=shape).astype(dtype)


@pytest.mark.parametrize("dtype", ["float64", "float32", "int32", "int64"])
def test_convolve_correlate_nd(gpu, sig, dtype):
    rng = np.random.default_rng(160)
    tol = dict(rtol=2e-6, atol=2e-5) if dtype == "float32" else dict(rtol=1e-12, atol=1e-12)
    cases = [((40,), (5,)), ((40,), (6,)), ((17, 23), (3, 3)), ((17, 23), (4, 5)), ((17, 23), (2, 1)), ((9, 10, 11), (3, 2, 4)),
             ((6,), (9,)), ((5, 6), (7, 8))]
    for s1, s2 in cases:
        a, b = _rand(rng, s1, dtype), _rand(rng, s2, dtype)
        for mode in ("full", "same", "valid"):
            for name in ("convolve", "correlate"):
                want = getattr(ssig, name)(a, b, mode=mode, method="direct")
                got = getattr(sig, name)(gpu.asarray(a), gpu.asarray(b), mode=mode)
                assert got.shape == want.shape, (name, s1, s2, mode)
                assert got.dtype == want.dtype, (name, s1, s2, mode)
                if np.dtype(dtype).kind == "f":
                    np.testing.assert_allclose(got.get(), want, **tol)
                else:
                    assert np.array_equal(got.get(), want), (name, s1, s2, mode)
    # mixed dtypes promote like NumPy; host weights are accepted
    a, b = _rand(rng, (20, 21), "int32"), _rand(rng, (3, 3), "float32")
    got = sig.convolve(gpu.asarray(a), b, mode="same")
    want = ssig.convolve(a, b, mode="same", method="direct")
    assert got.dtype == want.dtype
    np.testing.assert_allclose(got.get(), want, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("dtype", ["float64", "int32"])
def test_convolve2d_correlate2d(gpu, sig, dtype):
    rng = np.random.default_rng(161)
    for s1, s2 in [((16, 19), (3, 3)), ((16, 19), (4, 2)), ((16, 19), (1, 5)), ((12, 9), (5, 6)), ((4, 5), (6, 7))]:
        a, b = _rand(rng, s1, dtype), _rand(rng, s2, dtype)
        for mode in ("full", "same", "valid"):
            for boundary in ("fill", "wrap", "symm"):
                if boundary != "fill" and any(k > n for k, n in zip(s2, s1)):
                    continue        # padding wider than the image: np.pad semantics differ, not on the path
                for fv in ((0, 3) if boundary == "fill" else (0,)):
                    for name in ("convolve2d", "correlate2d"):
                        want = getattr(ssig, name)(a, b, mode=mode, boundary=boundary, fillvalue=fv)
                        got = getattr(sig, name)(gpu.asarray(a), gpu.asarray(b), mode=mode, boundary=boundary, fillvalue=fv)
                        assert got.shape == want.shape and got.dtype == want.dtype, (name, s1, s2, mode, boundary)
                        np.testing.assert_allclose(got.get(), want, rtol=1e-12, atol=1e-12,
                                                   err_msg=str((name, s1, s2, mode, boundary, fv)))


def test_signal_errors(gpu, sig):
    a = gpu.zeros((5, 5), np.float64)
    with pytest.raises(ValueError):
        sig.convolve(a, np.zeros(3))
    with pytest.raises(ValueError):
        sig.convolve(a, np.zeros((3, 3)), mode="bogus")
    with pytest.raises(ValueError):
        sig.convolve(a, np.zeros((3, 7)), mode="valid")
    with pytest.raises(NotImplementedError):
        sig.convolve(a, np.zeros((3, 3)), method="fft")
    with pytest.raises(ValueError):
        sig.convolve2d(a, np.zeros((3, 3)), boundary="bogus")
    with pytest.raises(ValueError):
        sig.convolve2d(gpu.zeros((5,), np.float64), np.zeros(3))
    from cupyimg_amd.scipy import ndimage as ndi
    with pytest.raises(ValueError):
        ndi.convolve(a, np.ones((3, 3)), output=gpu.zeros((5, 5), np.float64), dtype_mode="numpy")


def test_pad_matches_numpy(gpu):
    from cupyimg_amd import _pad
    rng = np.random.default_rng(162)
    for dtype in ("uint8", "int16", "float32", "float64", "bool"):
        x = rng.random((7, 9)) > 0.5 if dtype == "bool" else (rng.random((7, 9)) * 100).astype(dtype)
        for mode in ("constant", "edge", "wrap", "symmetric", "reflect"):
            for pw in ([(2, 3), (0, 4)], [(1, 1), (1, 1)], 2):
                kw = {"constant_values": 7} if mode == "constant" and dtype != "bool" else {}
                assert np.array_equal(_pad.pad(gpu.asarray(x), pw, mode, **kw).get(), np.pad(x, pw, mode=mode, **kw)), (dtype, mode, pw)
    v = rng.random((4, 5, 6)).astype(np.float32)
    assert np.array_equal(_pad.pad(gpu.asarray(v), [(1, 0), (0, 2), (3, 3)], "symmetric").get(),
                          np.pad(v, [(1, 0), (0, 2), (3, 3)], mode="symmetric"))


def test_convolve_separable(gpu):
    import cupyimg_amd as ca
    rng = np.random.default_rng(163)
    x = rng.standard_normal((12, 14, 10))
    w = rng.standard_normal(5)
    want = x
    for ax in range(3):
        want = sndi.convolve1d(want, w, axis=ax)
    np.testing.assert_allclose(ca.convolve_separable(gpu.asarray(x), w).get(), want, rtol=1e-12, atol=1e-12)
    ws = [rng.standard_normal(3), rng.standard_normal(4)]
    want = sndi.convolve1d(sndi.convolve1d(x, ws[0], axis=0, mode="nearest"), ws[1], axis=2, mode="nearest")
    got = ca.convolve_separable(gpu.asarray(x), [gpu.asarray(ws[0]), ws[1]], axes=(0, 2), mode="nearest")
    np.testing.assert_allclose(got.get(), want, rtol=1e-12, atol=1e-12)
    # float32 volumes with odd kernels: all passes in one fused launch (float32 accumulation, like convolve1d's default)
    v = rng.standard_normal((40, 36, 264)).astype(np.float32)
    for ws_, axes, kw in [([rng.standard_normal(5)] * 3, None, {}), ([rng.standard_normal(3), rng.standard_normal(9)], (2, 0), {"mode": "mirror"}),
                          ([rng.standard_normal(13)], (1,), {"mode": "wrap", "origin": 2}), ([rng.standard_normal(7)] * 3, None, {"mode": "constant", "cval": 0.5})]:
        wl = ws_[0] if axes is None else ws_
        want = v.astype(np.float64)
        for ax, w0 in zip(range(3) if axes is None else axes, ws_ if axes is not None else ws_):
            want = sndi.convolve1d(want, w0, axis=ax, **kw)
        got = ca.convolve_separable(gpu.asarray(v), wl, axes=axes, **kw)
        assert got.dtype == np.float32
        assert np.abs(got.get() - want).max() <= 2e-6 * np.abs(want).max(), (axes, kw)
    with pytest.raises(ValueError):
        ca.convolve_separable(gpu.asarray(x), [w], axes=(0, 1))
    with pytest.raises(ValueError):
        ca.convolve_separable(gpu.asarray(x), w, axes=(3,))
    with pytest.raises(ValueError):
        ca.convolve_separable(gpu.asarray(x), np.ones((2, 2)))
