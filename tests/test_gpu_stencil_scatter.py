"""Dense 3 x 3 x 3 / 5 x 5 x 5 (/ 7 x 7 x 7 in float mode) correlate / convolve on float32 volumes through stencil3s_kernel (csrc/stencil3s.hip;
reference filters.py:65-210, dtype_mode :470-487): the default mode (float64 accumulation in window order) is bit-identical
to the LDS-ring kernel, the generic kernel and SciPy's correlate on float64 input rounded to float32; dtype_mode="float"
(float32 accumulation, FMA) is within 1e-6 max-norm of SciPy.  Boundary modes, origins along z / y, ragged tile edges,
chunk seams, convolve (mirrored weights, negated origins), windows the kernel must refuse (zero weights, x origin)."""
import ctypes

import numpy as np
import pytest
import scipy.ndimage as sndi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ndi(gpu):
    from cupyimg_amd.scipy import ndimage
    return ndimage


@pytest.fixture()
def knob(gpu):
    from cupyimg_amd import _lib
    fn = _lib.load().mi_debug_set_stencil_scatter
    fn.argtypes = [ctypes.c_int]
    yield fn
    fn(1)


def maxnorm_rel(got, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(got, np.float64) - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize("W", [3, 5])
@pytest.mark.parametrize("shape", [(40, 48, 256), (19, 37, 520), (64, 130, 72), (9, 9, 1032)])
def test_scatter_correlate_modes_origins(gpu, ndi, knob, W, shape):
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(W * 1000 + shape[2])
    x = rng.standard_normal(shape).astype(np.float32)
    xd = gpu.asarray(x)
    w = rng.standard_normal((W, W, W))
    for mode, cval, origin in [("reflect", 0.0, 0), ("constant", 1.5, 0), ("nearest", 0.0, (1, -1, 0)), ("mirror", 0.0, (-1, 0, 0)),
                               ("wrap", 0.0, (0, 1, 0)), ("constant", 0.0, (W // 2, -(W // 2), 0))]:
        for fn, sfn in [(ndi.correlate, sndi.correlate), (ndi.convolve, sndi.convolve)]:
            knob(1)
            got = fn(xd, w, mode=mode, cval=cval, origin=origin)
            assert ("stencil3s_kernel<%d,double" % W in last_kernel()) == (W == 3), last_kernel()   # 5^3 float64 weights: ring kernel
            got = got.get()
            knob(0)
            ring = fn(xd, w, mode=mode, cval=cval, origin=origin).get()
            assert np.array_equal(got, ring), (fn.__name__, mode, origin, int((got != ring).sum()))
            ref = sfn(x.astype(np.float64), w, mode=mode, cval=cval, origin=origin).astype(np.float32)
            assert np.array_equal(got, ref), (fn.__name__, mode, origin)
            knob(1)
            gotf = fn(xd, w, mode=mode, cval=cval, origin=origin, dtype_mode="float")
            assert "stencil3s_kernel<%d,float" % W in last_kernel(), last_kernel()
            assert maxnorm_rel(gotf.get(), sfn(x.astype(np.float64), w, mode=mode, cval=cval, origin=origin)) <= 1e-6, (mode, origin)


@pytest.mark.parametrize("W", [3, 5])
def test_scatter_float32_valued_weights_use_fma_and_stay_exact(gpu, ndi, knob, W):
    """Weights that are float32 values (a float32 kernel, small integers): sample x weight is exact in float64, the default
    mode runs on v_fma_f64 and must still be bit-identical to SciPy and to the mul + add kernels."""
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(50 + W)
    x = (rng.standard_normal((33, 50, 264)) * 1e3).astype(np.float32)
    xd = gpu.asarray(x)
    for w in (rng.standard_normal((W, W, W)).astype(np.float32), rng.integers(1, 9, size=(W, W, W)).astype(np.float64),
              np.full((W, W, W), 0.125)):
        got = ndi.correlate(xd, w, mode="mirror")
        if W == 3:                                           # 5^3: from 2^25 voxels (test_scatter_full_size_512_every_plane)
            assert "stencil3s_kernel<3,double" in last_kernel() and "fma" in last_kernel(), last_kernel()
        got = got.get()
        knob(0)
        ring = ndi.correlate(xd, w, mode="mirror").get()
        knob(1)
        assert np.array_equal(got, ring)
        assert np.array_equal(got, sndi.correlate(x.astype(np.float64), np.asarray(w, np.float64), mode="mirror").astype(np.float32))
    w = rng.standard_normal((W, W, W))                       # float64 weights: products round, mul + add stays
    ndi.correlate(xd, w)
    assert ("mul + add" in last_kernel()) if W == 3 else ("stencil3_kernel<" in last_kernel()), last_kernel()


def test_scatter_refusals_fall_back(gpu, ndi, knob):
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(9)
    x = rng.standard_normal((32, 40, 128)).astype(np.float32)
    xd = gpu.asarray(x)
    w = rng.standard_normal((3, 3, 3))
    w0 = w.copy()
    w0[1, 0, 2] = 0.0                                        # a zero weight is skipped by the reference: inf * 0 must not appear
    xi = x.copy()
    xi[10, 10, 10] = np.inf
    got = ndi.correlate(gpu.asarray(xi), w0).get()
    assert "stencil3s_kernel" not in last_kernel()
    ref = sndi.correlate(xi.astype(np.float64), w0).astype(np.float32)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(got[np.isfinite(ref)], ref[np.isfinite(ref)])
    got = ndi.correlate(xd, w, origin=(0, 0, 1)).get()       # x origin: the ring kernel's padded window
    assert "stencil3s_kernel" not in last_kernel()
    assert np.array_equal(got, sndi.correlate(x.astype(np.float64), w, origin=(0, 0, 1)).astype(np.float32))
    # non-finite samples with a dense window propagate like SciPy's
    got = ndi.correlate(gpu.asarray(xi), w).get()
    assert "stencil3s_kernel<3,double" in last_kernel()
    ref = sndi.correlate(xi.astype(np.float64), w).astype(np.float32)
    fin = np.isfinite(ref)
    assert np.array_equal(fin, np.isfinite(got)) and np.array_equal(got[fin], ref[fin])


@pytest.mark.parametrize("W", [3, 5])
def test_scatter_full_size_512_every_plane(gpu, ndi, W):
    """512^3: every plane of the last launch of a burst against SciPy (float64 correlate rounded to float32: exact)."""
    from helpers import fullsize as fs
    from cupyimg_amd import last_kernel
    gpu.free_all_blocks()
    x = fs.volume_f32((512,) * 3, seed=0)
    xd = gpu.asarray(x)
    w = np.random.default_rng(W).standard_normal((W, W, W))
    out = gpu.empty(x.shape, np.float32)
    for _ in range(6):
        ndi.correlate(xd, w, output=out)
    assert ("stencil3s_kernel<%d,double" % W in last_kernel()) == (W == 3), last_kernel()
    bad = fs.whole_volume_filter(x, out.get(), W // 2, W // 2, lambda s: sndi.correlate(s.astype(np.float64), w).astype(np.float32),
                                 exact=True, planes=8)
    assert bad == 0, bad
    w32 = w.astype(np.float32)                              # float32-valued weights: v_fma_f64 with exact products
    for _ in range(6):
        ndi.correlate(xd, w32, output=out)
    assert "stencil3s_kernel<%d,double" % W in last_kernel() and "fma" in last_kernel(), last_kernel()
    bad = fs.whole_volume_filter(x, out.get(), W // 2, W // 2,
                                 lambda s: sndi.correlate(s.astype(np.float64), w32.astype(np.float64)).astype(np.float32), exact=True, planes=8)
    assert bad == 0, bad
    for _ in range(6):
        ndi.correlate(xd, w, output=out, dtype_mode="float")
    assert "stencil3s_kernel<%d,float" % W in last_kernel(), last_kernel()
    err = fs.whole_volume_filter(x, out.get(), W // 2, W // 2, lambda s: sndi.correlate(s.astype(np.float64), w), planes=8)
    assert err <= 1e-6, err
    del xd, out
    gpu.free_all_blocks()


@pytest.mark.parametrize("shape", [(40, 48, 256), (19, 37, 520), (64, 130, 72), (9, 9, 1032)])
def test_scatter_7x7x7_float_mode(gpu, ndi, knob, shape):
    """7 x 7 x 7 in dtype_mode="float": the lane-weight tap rows (343 weights in six registers); the default mode stays on the
    LDS-ring kernel (FP64 pipe bound either way).  Tolerance 2e-6 of the max-norm: 343 float32 products summed in float32 -- what
    the reference's own float mode does (filters.py:470-487) -- sit at 7-9e-7 of SciPy's float64 sum on these volumes."""
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(7000 + shape[2])
    x = rng.standard_normal(shape).astype(np.float32)
    xd = gpu.asarray(x)
    w = rng.standard_normal((7, 7, 7))
    for mode, cval, origin in [("reflect", 0.0, 0), ("constant", 1.5, 0), ("nearest", 0.0, (1, -1, 0)), ("mirror", 0.0, (-3, 0, 0)),
                               ("wrap", 0.0, (0, 3, 0)), ("constant", 0.0, (3, -3, 0))]:
        for fn, sfn in [(ndi.correlate, sndi.correlate), (ndi.convolve, sndi.convolve)]:
            knob(1)
            gotf = fn(xd, w, mode=mode, cval=cval, origin=origin, dtype_mode="float")
            assert "stencil3s_kernel<7,float" in last_kernel(), last_kernel()
            ref = sfn(x.astype(np.float64), w, mode=mode, cval=cval, origin=origin)
            assert maxnorm_rel(gotf.get(), ref) <= 2e-6, (fn.__name__, mode, origin)
            got = fn(xd, w, mode=mode, cval=cval, origin=origin)
            assert "stencil3s_kernel" not in last_kernel(), last_kernel()
            assert np.array_equal(got.get(), ref.astype(np.float32)), (fn.__name__, mode, origin)
    # every weight distinct and in its place: a delta volume returns the mirrored window
    d = np.zeros((21, 21, 256), np.float32)
    d[10, 10, 100] = 1.0
    w32 = rng.standard_normal((7, 7, 7)).astype(np.float32)
    got = ndi.correlate(gpu.asarray(d), w32, dtype_mode="float").get()
    assert "stencil3s_kernel<7,float" in last_kernel(), last_kernel()
    assert np.array_equal(got[7:14, 7:14, 97:104], w32[::-1, ::-1, ::-1])


def test_scatter_7x7x7_full_size_512(gpu, ndi):
    from helpers import fullsize as fs
    from cupyimg_amd import last_kernel
    gpu.free_all_blocks()
    x = fs.volume_f32((512,) * 3, seed=0)
    xd = gpu.asarray(x)
    w = np.random.default_rng(7).standard_normal((7, 7, 7))
    out = gpu.empty(x.shape, np.float32)
    for _ in range(4):
        ndi.correlate(xd, w, output=out, dtype_mode="float")
    assert "stencil3s_kernel<7,float" in last_kernel(), last_kernel()
    err = fs.whole_volume_filter(x, out.get(), 3, 3, lambda s: sndi.correlate(s.astype(np.float64), w), planes=8)
    assert err <= 2e-6, err                                 # 343 float32 terms: see test_scatter_7x7x7_float_mode
    del xd, out
    gpu.free_all_blocks()


@pytest.mark.parametrize("W", [3, 5, 7])
def test_scatter_rows_of_any_length(gpu, ndi, knob, W):
    """Rows that are not a multiple of four floats (r6, late): the scatter kernel stages them as they lie (mi_correlate3_dense, asked
    before the rows are extended for the tiled kernel); one tile and two tiles per row, every tail, every mode; what the kernel
    refuses on such rows (zero weights, x origins) still gives SciPy's numbers through the extended-rows route."""
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(4400 + W)
    tails = set()
    for shape in [(24, 37, 181), (17, 30, 301), (12, 40, 253), (20, 21, 255), (16, 19, 257), (70, 64, 18), (64, 33, 35), (9, 20, 511),
                  (181, 217, 181)]:
        x = rng.standard_normal(shape).astype(np.float32)
        xd = gpu.asarray(x)
        w = rng.standard_normal((W, W, W))
        modes = [("reflect", 0.0, 0), ("constant", 1.5, 0), ("nearest", 0.0, (1, -1, 0)), ("mirror", 0.0, 0), ("wrap", 0.0, (0, 1, 0))]
        for mode, cval, origin in (modes if shape[0] < 100 else modes[:2]):
            for fn, sfn in [(ndi.correlate, sndi.correlate), (ndi.convolve, sndi.convolve)]:
                ref = sfn(x.astype(np.float64), w, mode=mode, cval=cval, origin=origin)
                gotf = fn(xd, w, mode=mode, cval=cval, origin=origin, dtype_mode="float")
                assert "stencil3s_kernel<%d,float" % W in last_kernel(), (shape, last_kernel())
                assert maxnorm_rel(gotf.get(), ref) <= (2e-6 if W == 7 else 1e-6), (shape, mode, origin)
                if W == 3:
                    got = fn(xd, w, mode=mode, cval=cval, origin=origin)
                    assert "stencil3s_kernel<3,double" in last_kernel(), (shape, last_kernel())
                    assert np.array_equal(got.get(), ref.astype(np.float32)), (shape, mode, origin)
        tails.add(shape[2] & 3)
        w0 = w.copy()
        w0[0, 1, 1] = 0.0
        assert np.array_equal(ndi.correlate(xd, w0).get(), sndi.correlate(x.astype(np.float64), w0).astype(np.float32)), shape
        assert "stencil3s_kernel" not in last_kernel()
        got = ndi.correlate(xd, w, origin=(0, 0, 1), dtype_mode="float").get()
        assert maxnorm_rel(got, sndi.correlate(x.astype(np.float64), w, origin=(0, 0, 1))) <= 2e-6, shape
    assert tails == {1, 2, 3}
    # an output the caller provides, a view that starts inside a buffer (4-byte aligned only)
    big = gpu.asarray(rng.standard_normal((40, 50, 77)).astype(np.float32))
    sub = big[3:35]
    out = gpu.empty(sub.shape, np.float32)
    w = rng.standard_normal((W, W, W))
    ndi.correlate(sub, w, output=out, dtype_mode="float")
    assert "stencil3s_kernel" in last_kernel()
    assert maxnorm_rel(out.get(), sndi.correlate(big.get()[3:35].astype(np.float64), w)) <= 2e-6
