"""r5 additions to the filter kernels.  Rows that are not a multiple of four floats go through the 3 / 5 / 7-tap fused kernel as they are
(sep3d_lean_kernel<..., ragged>: csrc/separable3d.hip) -- no mi_extend_rows / mi_crop_rows copies around the launch -- and
rank filters with 65 .. 128 samples (5 x 5 x 5, 9 x 9, 11 x 11) take the register sorting network (rank_sorted_p128*.hip)
instead of the scratch-array selection kernel; `constant` mode with a zero fill value on the LDS-DMA kernel of 9 .. 17 taps
(sep3d_long3_kernel: zero fill is what its staging leaves for lanes beyond the array).  Spec: /root/reference/cupyimg/scipy/ndimage/filters.py:549-665 (separable
passes), :1560-1850 (rank / median / percentile filters)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODES = ("reflect", "mirror", "nearest", "wrap", "constant")


@pytest.fixture(scope="module")
def ndi(gpu):
    from cupyimg_amd.scipy import ndimage
    return ndimage


@pytest.fixture(scope="module")
def lib(gpu):
    from cupyimg_amd import _lib
    return _lib.load()


def _shapes():
    # last lane holds 1, 2, 3 floats; one and two x tiles; rows shorter than a wave's 256 floats and just beyond; ny, nz that
    # leave partial y tiles and one-plane chunks
    return [(24, 37, 181), (17, 30, 301), (9, 40, 253), (9, 21, 255), (12, 19, 257), (10, 18, 17), (8, 33, 19), (5, 20, 511),
            (40, 7, 66), (3, 3, 18), (1, 64, 129)]


@pytest.mark.parametrize("taps", [3, 5, 7, 9])
def test_ragged_rows_every_mode_against_scipy_and_the_extended_rows_route(gpu, ndi, lib, taps):
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(500 + taps)
    seen_tails = set()
    for shape in _shapes():
        x = rng.standard_normal(shape).astype(np.float32)
        xd = gpu.asarray(x)
        for mode in MODES:
            kw = dict(mode=mode, cval=-0.75)
            got = ndi.uniform_filter(xd, taps, **kw).get()
            k = last_kernel()
            if taps == 9 and mode != "constant":          # r6: nine taps with one weight vector take the LDS-DMA kernel's ragged build (a fill value keeps the lean one)
                assert "sep3d_long3_kernel<9,true,ragged>" in k, (shape, mode, k)
            else:
                assert "ragged" in k and "sep3d_lean_kernel<%d," % taps in k, (shape, mode, k)
            ref = sndi.uniform_filter(x.astype(np.float64), taps, **kw)
            assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (shape, mode, k)
            seen_tails.add(shape[2] & 3)
            # the same kernel on explicitly extended rows (the r4b route): the same sums in the same order
            if x.size >= (1 << 15):
                lib.mi_debug_set_sep3d_ragged(0)
                try:
                    via = ndi.uniform_filter(xd, taps, **kw).get()
                    kv = last_kernel()
                finally:
                    lib.mi_debug_set_sep3d_ragged(1)
                assert "ragged" not in kv, kv
                if "sep3d_lean_kernel" in kv and taps <= 7:
                    assert np.array_equal(got, via), (shape, mode, k, kv)
                else:
                    assert np.abs(got - via).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (shape, mode, k, kv)
        # a gaussian of the same tap count (weights that are not all equal: the order of the continuation matters)
        sigma = {3: 0.25, 5: 0.5, 7: 0.75, 9: 1.0}[taps]
        for mode in MODES:
            got = ndi.gaussian_filter(xd, sigma, mode=mode, cval=2.5).get()
            assert "ragged" in last_kernel(), (shape, mode, last_kernel())
            ref = sndi.gaussian_filter(x.astype(np.float64), sigma, mode=mode, cval=2.5)
            assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (shape, mode, sigma)
    assert seen_tails == {1, 2, 3}


def test_ragged_rows_non_finite_values_and_caller_outputs(gpu, ndi, lib):
    """NaN / inf next to the row ends (they spread over the window and no further), output into a caller's array, rows one,
    two and three floats past a multiple of four on either side of the 256-float tile width"""
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(77)
    for nx in (181, 182, 183, 185, 253, 254, 255, 257, 258, 259):
        x = rng.standard_normal((11, 23, nx)).astype(np.float32)
        x[3, 5, nx - 1] = np.inf
        x[4, 6, nx - 2] = np.nan
        x[5, 7, 0] = -np.inf
        xd = gpu.asarray(x)
        out = gpu.empty(x.shape, np.float32)
        for taps in (3, 5, 7):
            for mode in MODES:
                r = ndi.uniform_filter(xd, taps, mode=mode, cval=0.5, output=out)
                assert r is out and "ragged" in last_kernel(), last_kernel()
                g = out.get()
                # SciPy's uniform_filter1d is a running sum (a NaN poisons the rest of its line): the reference for where the
                # non-finite values land is the same kernel on explicitly extended rows
                lib.mi_debug_set_sep3d_ragged(0)
                try:
                    via = ndi.uniform_filter(xd, taps, mode=mode, cval=0.5).get()
                    kv = last_kernel()
                finally:
                    lib.mi_debug_set_sep3d_ragged(1)
                assert "ragged" not in kv and "sep3d_lean_kernel" in kv, kv
                assert np.array_equal(g, via, equal_nan=True), (nx, taps, mode)
                assert 0 < np.isnan(g).sum() <= taps ** 3 + (taps ** 3 if mode == "wrap" else 0), (nx, taps, mode, int(np.isnan(g).sum()))
                xf = np.where(np.isfinite(x), x, 0.0).astype(np.float64)
                want = sndi.uniform_filter(xf, taps, mode=mode, cval=0.5)
                fin = np.isfinite(g)
                # finite outputs saw no non-finite input: they agree with SciPy on the volume with those voxels zeroed
                assert np.abs(g[fin] - want[fin]).max() <= 1e-6 * max(1.0, np.abs(want).max()), (nx, taps, mode)


def test_ragged_rows_whole_mni_volume_last_of_a_burst(gpu, ndi, lib):
    """181 x 217 x 181 (the MNI152 1 mm grid): the last launch of a burst, every plane against SciPy"""
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(9)
    x = rng.standard_normal((181, 217, 181)).astype(np.float32)
    xd = gpu.asarray(x)
    out = gpu.empty(x.shape, np.float32)
    for size, mode in ((5, "reflect"), (3, "mirror"), (7, "constant")):
        for _ in range(30):
            ndi.uniform_filter(xd, size, mode=mode, cval=1.5, output=out)
        assert "ragged" in last_kernel(), last_kernel()
        ref = sndi.uniform_filter(x.astype(np.float64), size, mode=mode, cval=1.5)
        g = out.get()
        err = np.abs(g - ref).reshape(181, -1).max(axis=1)
        assert err.max() <= 1e-6 * np.abs(ref).max(), (size, mode, int(err.argmax()), float(err.max()))


def test_rank_filters_of_65_to_128_samples_take_the_sorting_network(gpu, ndi, lib):
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(4242)
    vol = (rng.standard_normal((14, 19, 70)) * 60 + 100)
    img = (rng.standard_normal((61, 135)) * 60 + 100)
    for dt in (np.float32, np.uint8, np.int16, np.uint16, np.int8, np.int32, np.uint32):
        for x in (vol, img):
            x = np.clip(x, np.iinfo(dt).min, np.iinfo(dt).max).astype(dt) if np.dtype(dt).kind in "iu" else x.astype(dt)
            xd = gpu.asarray(x)
            sizes = [5] if x.ndim == 3 else [9, 11]
            for size in sizes:
                n = size ** x.ndim
                for mode in ("reflect", "constant", "wrap"):
                    got = ndi.median_filter(xd, size=size, mode=mode, cval=3).get()
                    assert "rank3_sorted_kernel" in last_kernel() and ",128," in last_kernel(), last_kernel()
                    assert np.array_equal(got, sndi.median_filter(x, size=size, mode=mode, cval=3)), (dt, size, mode)
                for rank in (0, 1, n // 3, n - 2, n - 1):
                    got = ndi.rank_filter(xd, rank, size=size, mode="mirror").get()
                    assert np.array_equal(got, sndi.rank_filter(x, rank, size=size, mode="mirror")), (dt, size, rank)
                got = ndi.percentile_filter(xd, 30, size=size, mode="nearest").get()
                assert np.array_equal(got, sndi.percentile_filter(x, 30, size=size, mode="nearest")), (dt, size)
            # footprints with holes: 65 and 128 samples exactly, an origin
            shape_fp = (5, 5, 6) if x.ndim == 3 else (10, 13)
            for nset in (65, 100, 128):
                fp = np.zeros(int(np.prod(shape_fp)), bool)
                fp[rng.permutation(fp.size)[:nset]] = True
                fp = fp.reshape(shape_fp)
                org = (1, -1, 0) if x.ndim == 3 else (-2, 3)
                got = ndi.rank_filter(xd, nset // 2, footprint=fp, origin=org, mode="reflect").get()
                assert "rank3_sorted_kernel" in last_kernel(), last_kernel()
                assert np.array_equal(got, sndi.rank_filter(x, nset // 2, footprint=fp, origin=org, mode="reflect")), (dt, nset)
    # infinities sort like any other value
    x = vol.astype(np.float32)
    x[3, 4, 5] = np.inf
    x[7, 8, 9] = -np.inf
    assert np.array_equal(ndi.median_filter(gpu.asarray(x), size=5).get(), sndi.median_filter(x, size=5))
    # float64 keeps the selection kernel beyond 64 samples
    xd64 = gpu.asarray(vol)
    got = ndi.median_filter(xd64, size=5).get()
    assert "rank3_sorted_kernel" not in last_kernel()
    assert np.array_equal(got, sndi.median_filter(vol, size=5))


def test_rank_network_orders_nans_and_signed_zeros_like_numpy_sort(gpu, ndi, lib):
    """r5: the sorting network works on integer keys (a total order): NaNs (sign bit clear, what arithmetic produces) sort above
    +inf -- where numpy.sort puts them, so a rank filter equals sorted(window)[rank] -- and -0.0 below +0.0.  SciPy's own
    selection leaves the result with NaNs to the order of its comparisons; the reference's kernels likewise."""
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(99)
    x = rng.standard_normal((6, 9, 70)).astype(np.float32)
    x[rng.random(x.shape) < 0.05] = np.nan
    x[rng.random(x.shape) < 0.03] = np.inf
    x[rng.random(x.shape) < 0.03] = -np.inf
    x[rng.random(x.shape) < 0.05] = 0.0
    x[rng.random(x.shape) < 0.05] = -0.0
    xd = gpu.asarray(x)
    # (ranks 0 and n - 1 are minimum_filter / maximum_filter calls, as in the reference: not this kernel)
    for size, ranks in ((3, (1, 8, 13, 20, 25)), ((1, 3, 3), (1, 4, 7)), ((1, 5, 5), (12,)), (5, (62, 100, 123))):
        n = int(np.prod(size)) if isinstance(size, tuple) else size ** 3
        for rank in ranks:
            got = ndi.rank_filter(xd, rank, size=size, mode="reflect").get()
            assert "rank3_sorted_kernel" in last_kernel() or n in (25, 27), last_kernel()
            want = sndi.generic_filter(x.astype(np.float64), lambda w: np.sort(w)[rank], size=size, mode="reflect").astype(np.float32)
            assert np.array_equal(got, want, equal_nan=True), (size, rank)
    got = ndi.median_filter(xd, size=3).get()
    want = sndi.generic_filter(x.astype(np.float64), lambda w: np.sort(w)[13], size=3, mode="reflect").astype(np.float32)
    assert np.array_equal(got, want, equal_nan=True)


def test_constant_mode_with_zero_fill_on_the_long_kernel(gpu, ndi, lib):
    """mode="constant", cval=0 (SciPy's default fill): one launch of sep3d_long3_kernel -- rows, halo columns and planes
    beyond the array are not fetched -- against SciPy and against the r2 kernel with its coverage correction; mixed modes per
    axis, plane ranges through partial tiles, the anisotropic (W, WZ) pairs; any other fill value keeps the r2 kernel."""
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(808)
    for shape in ((40, 50, 256), (37, 45, 300), (19, 33, 64), (70, 18, 516)):
        x = rng.standard_normal(shape).astype(np.float32)
        xd = gpu.asarray(x)
        x64 = x.astype(np.float64)
        for size in (9, 13, 17):
            got = ndi.uniform_filter(xd, size, mode="constant").get()
            k = last_kernel()
            assert "sep3d_long3_kernel<%d," % size in k and "zero fill" in k, k
            ref = sndi.uniform_filter(x64, size, mode="constant")
            assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (shape, size)
            lib.mi_debug_set_long_const0(0)
            try:
                old = ndi.uniform_filter(xd, size, mode="constant").get()
                assert "sep3d_long_kernel<%d," % size in last_kernel(), last_kernel()
            finally:
                lib.mi_debug_set_long_const0(1)
            assert np.abs(got - old).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (shape, size)
            # a fill value: the r2 kernel
            got = ndi.uniform_filter(xd, size, mode="constant", cval=1.5).get()
            assert "sep3d_long_kernel<%d," % size in last_kernel(), last_kernel()
            assert np.abs(got - sndi.uniform_filter(x64, size, mode="constant", cval=1.5)).max() <= 1e-6 * 4
        for sigma in (1.0, 1.5, 2.0):
            for modes in ("constant", ("constant", "reflect", "mirror"), ("nearest", "constant", "wrap"), ("wrap", "mirror", "constant")):
                got = ndi.gaussian_filter(xd, sigma, mode=modes).get()
                assert "sep3d_long3_kernel" in last_kernel(), (shape, sigma, modes, last_kernel())
                ref = sndi.gaussian_filter(x64, sigma, mode=modes)
                assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (shape, sigma, modes)
    # anisotropic voxels: fewer taps along z than in the plane (one launch for the pairs the kernel is built for)
    x = rng.standard_normal((64, 256, 260)).astype(np.float32)
    xd = gpu.asarray(x)
    got = ndi.gaussian_filter(xd, (1.0, 2.0, 2.0), mode="constant").get()
    assert "sep3d_long3_kernel<17,false,false,0,9>" in last_kernel(), last_kernel()
    ref = sndi.gaussian_filter(x.astype(np.float64), (1.0, 2.0, 2.0), mode="constant")
    assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())


def test_constant_zero_fill_512_every_plane_last_of_a_burst(gpu, ndi, lib):
    """gaussian_filter(sigma=2, mode="constant") on 512^3: the last launch of a burst, every plane against SciPy"""
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    x = np.random.default_rng(31).standard_normal((512, 512, 512)).astype(np.float32)
    xd = gpu.asarray(x)
    out = gpu.empty(x.shape, np.float32)
    for _ in range(30):
        ndi.gaussian_filter(xd, 2.0, mode="constant", output=out)
    assert "sep3d_long3_kernel<17," in last_kernel() and "zero fill" in last_kernel(), last_kernel()
    ref = sndi.gaussian_filter(x, 2.0, mode="constant")
    err = np.abs(out.get() - ref).reshape(512, -1).max(axis=1)
    assert err.max() <= 1e-6 * np.abs(ref).max() + 2e-7, (int(err.argmax()), float(err.max()))


def test_median_3x3x3_shared_sort_kernel(gpu, ndi, lib):
    """median_filter(size=3) on volumes: median27_stream_kernel (csrc/median3d.hip: the window sorted along z, x, y in turn,
    partial results shared between windows, a searched 19-input network at the end) -- bit-exact against SciPy and against the
    per-voxel network, every boundary mode, every dtype with 32-bit keys, tile edges, z chunks, NaNs as numpy.sort orders them."""
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(2727)
    shapes = [(40, 33, 70), (9, 200, 129), (130, 30, 62), (17, 15, 63), (33, 29, 64), (2, 128, 128), (64, 14, 125), (70, 16, 126)]
    for dt in (np.float32, np.uint8, np.int8, np.uint16, np.int16, np.int32, np.uint32, np.float64):
        for shape in shapes if dt in (np.float32, np.uint8, np.float64) else shapes[:3]:
            x = rng.standard_normal(shape) * 60 + 100
            x = np.clip(x, np.iinfo(dt).min, np.iinfo(dt).max).astype(dt) if np.dtype(dt).kind in "iu" else x.astype(dt)
            if dt == np.uint32:
                x = x * np.uint32(30000000)                     # beyond 2^31: unsigned keys
            if dt == np.int32:
                x = (x - 100) * np.int32(20000000)
            xd = gpu.asarray(x)
            for mode in MODES:
                got = ndi.median_filter(xd, size=3, mode=mode, cval=7).get()
                assert "median27_stream_kernel" in last_kernel(), (dt, shape, last_kernel())
                assert np.array_equal(got, sndi.median_filter(x, size=3, mode=mode, cval=7)), (dt, shape, mode)
            # the same through the footprint argument and percentile_filter; the per-voxel network agrees
            fp = np.ones((3, 3, 3), bool)
            got = ndi.percentile_filter(xd, 50, footprint=fp).get()
            assert "median27_stream_kernel" in last_kernel(), last_kernel()
            lib.mi_debug_set_median27(0)
            try:
                old = ndi.median_filter(xd, size=3).get()
                assert "median27_stream_kernel" not in last_kernel()
            finally:
                lib.mi_debug_set_median27(1)
            assert np.array_equal(got, old), (dt, shape)
    # an origin, a footprint with a hole, another rank: not this kernel
    x = rng.standard_normal((20, 40, 64)).astype(np.float32)
    xd = gpu.asarray(x)
    for kw in (dict(size=3, origin=(0, 1, 0)), dict(footprint=np.arange(27).reshape(3, 3, 3) != 5), dict(size=(3, 3, 5))):
        got = ndi.median_filter(xd, **kw).get()
        assert "median27_stream_kernel" not in last_kernel(), (kw, last_kernel())
        assert np.array_equal(got, sndi.median_filter(x, **kw)), kw
    # NaN / inf: a total order, NaNs above +inf (numpy.sort)
    x = rng.standard_normal((12, 20, 70)).astype(np.float32)
    x[rng.random(x.shape) < 0.06] = np.nan
    x[rng.random(x.shape) < 0.04] = np.inf
    x[rng.random(x.shape) < 0.04] = -np.inf
    got = ndi.median_filter(gpu.asarray(np.tile(x, (4, 1, 1))), size=3).get()[:12]
    want = sndi.generic_filter(np.tile(x, (4, 1, 1)).astype(np.float64), lambda w: np.sort(w)[13], size=3, mode="reflect").astype(np.float32)[:12]
    assert np.array_equal(got[:11], want[:11], equal_nan=True)


def test_median_3x3x3_whole_volume_last_of_a_burst(gpu, ndi, lib):
    """256^3 float32 and 181 x 217 x 181 uint8: last launch of a burst, every voxel"""
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(14)
    for shape, dt in (((256, 256, 256), np.float32), ((181, 217, 181), np.uint8), ((150, 200, 250), np.float64)):
        x = (rng.standard_normal(shape) * 50 + 100).astype(dt)
        xd = gpu.asarray(x)
        out = gpu.empty(shape, dt)
        for _ in range(20):
            ndi.median_filter(xd, size=3, output=out)
        assert "median27_stream_kernel" in last_kernel(), last_kernel()
        assert np.array_equal(out.get(), sndi.median_filter(x, size=3)), (shape, dt)


def test_every_rank_of_the_3x3x3_window_on_the_shared_sort_kernel(gpu, ndi, lib):
    """rank_filter / percentile_filter with the full 3 x 3 x 3 footprint: ranks 1 .. 25 each have their own candidate set and
    searched network (median27_net.hpp: Rank27Net<KO, R>); every dtype with 32-bit keys and float64 take it for every rank.  Bit-exact against SciPy; ranks 0 and 26 are minimum / maximum filters."""
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(272727)
    for dt in (np.float32, np.uint8, np.int16, np.uint16, np.float64):
        for shape in ((23, 31, 70), (5, 64, 130)):
            x = (rng.standard_normal(shape) * 60 + 100)
            x = np.clip(x, np.iinfo(dt).min, np.iinfo(dt).max).astype(dt) if np.dtype(dt).kind in "iu" else x.astype(dt)
            xd = gpu.asarray(x)
            for rank in range(0, 27):
                mode = MODES[rank % 5]
                got = ndi.rank_filter(xd, rank, size=3, mode=mode, cval=5).get()
                k = last_kernel()
                if 1 <= rank <= 25:
                    assert "median27_stream_kernel<%d>" % rank in k, (dt, rank, k)
                assert np.array_equal(got, sndi.rank_filter(x, rank, size=3, mode=mode, cval=5)), (dt, shape, rank, mode)
            for pct in (10, 25, 75, 90):
                got = ndi.percentile_filter(xd, pct, size=3).get()
                assert "median27_stream_kernel" in last_kernel(), last_kernel()
                assert np.array_equal(got, sndi.percentile_filter(x, pct, size=3)), (dt, shape, pct)
            assert np.array_equal(ndi.rank_filter(xd, -5, size=3).get(), sndi.rank_filter(x, -5, size=3))
    # the remaining dtypes with 32-bit keys (uint32 beyond 2^31: compared unsigned)
    for dt, scale in ((np.int8, 1), (np.int32, 20000000), (np.uint32, 30000000)):
        x = np.clip(rng.standard_normal((20, 30, 70)) * 40 + 60, 0 if dt == np.uint32 else -100, 120).astype(dt) * dt(scale)
        xd = gpu.asarray(x)
        for rank in (2, 7, 13, 19, 24):
            got = ndi.rank_filter(xd, rank, size=3, mode="mirror").get()
            assert "median27_stream_kernel<%d>" % rank in last_kernel(), last_kernel()
            assert np.array_equal(got, sndi.rank_filter(x, rank, size=3, mode="mirror")), (dt, rank)
    # float64: infinities sort like any value (NaNs are passed over by v_min_f64 / v_max_f64: no contract)
    x = rng.standard_normal((9, 20, 70))
    x[rng.random(x.shape) < 0.06] = np.inf
    x[rng.random(x.shape) < 0.06] = -np.inf
    for rank in (3, 13, 22):
        assert np.array_equal(ndi.rank_filter(gpu.asarray(x), rank, size=3).get(), sndi.rank_filter(x, rank, size=3)), rank
    # NaNs: numpy.sort's order for every rank
    x = rng.standard_normal((8, 20, 70)).astype(np.float32)
    x[rng.random(x.shape) < 0.08] = np.nan
    x[rng.random(x.shape) < 0.05] = -np.inf
    xd = gpu.asarray(np.tile(x, (4, 1, 1)))
    for rank in (1, 6, 12, 14, 20, 25):
        got = ndi.rank_filter(xd, rank, size=3).get()[:7]
        want = sndi.generic_filter(np.tile(x, (4, 1, 1)).astype(np.float64), lambda w: np.sort(w)[rank], size=3, mode="reflect").astype(np.float32)[:7]
        assert np.array_equal(got, want, equal_nan=True), rank


def test_constant_zero_fill_below_nine_taps_on_large_volumes(gpu, ndi, lib):
    """3 / 5 / 7 taps, mode="constant", cval=0 on volumes large enough for the LDS-DMA kernel: the same kernel as the other
    modes; a fill value keeps the lean kernel.  Against SciPy, every plane."""
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    x = np.random.default_rng(55).standard_normal((160, 256, 256)).astype(np.float32)
    xd = gpu.asarray(x)
    x64 = x.astype(np.float64)
    for size in (3, 5, 7):
        got = ndi.uniform_filter(xd, size, mode="constant").get()
        assert "sep3d_long3_kernel<%d," % size in last_kernel() and "zero fill" in last_kernel(), last_kernel()
        ref = sndi.uniform_filter(x64, size, mode="constant")
        err = np.abs(got - ref).reshape(160, -1).max(axis=1)
        assert err.max() <= 1e-6 * np.abs(ref).max(), (size, int(err.argmax()))
        got = ndi.uniform_filter(xd, size, mode="constant", cval=2.0).get()
        assert "sep3d_lean_kernel" in last_kernel(), last_kernel()
        assert np.abs(got - sndi.uniform_filter(x64, size, mode="constant", cval=2.0)).max() <= 1e-6 * 4
    lib.mi_debug_set_long_const0(0)
    try:
        got = ndi.uniform_filter(xd, 5, mode="constant").get()
        assert "sep3d_lean_kernel" in last_kernel(), last_kernel()
    finally:
        lib.mi_debug_set_long_const0(1)
    assert np.abs(got - sndi.uniform_filter(x64, 5, mode="constant")).max() <= 1e-6 * 4


@pytest.mark.parametrize("size,mode,cval", [(3, "reflect", 0.0), (5, "constant", 0.5), (5, "wrap", 0.0), (7, "mirror", 0.0), (7, "constant", 0.0)])
def test_ragged_rows_plane_restricted_launches_tile_the_full_result(gpu, ndi, lib, size, mode, cval):
    """the slab path (output planes given by the caller: distributed.SlabFilter) on rows that are not a multiple of four
    floats: the 3 / 5 / 7-tap kernel takes them as they are, plane ranges included"""
    from cupyimg_amd import last_kernel
    from cupyimg_amd.scipy.ndimage import _support as S
    rng = np.random.default_rng(12)
    x = rng.standard_normal((70, 45, 301)).astype(np.float32)
    xd = gpu.asarray(x)
    full = ndi.uniform_filter(xd, size, mode=mode, cval=cval).get()
    assert "ragged" in last_kernel(), last_kernel()
    sentinel = np.float32(-12345.0)
    out = gpu.asarray(np.full(x.shape, sentinel, np.float32))
    with S.output_planes([(4, 31)]):
        ndi.uniform_filter(xd, size, mode=mode, cval=cval, output=out)
    assert "ragged" in last_kernel(), last_kernel()
    got = out.get()
    assert np.array_equal(got[4:31], full[4:31])
    assert np.all(got[:4] == sentinel) and np.all(got[31:] == sentinel)     # nothing else written
    with S.output_planes([(0, 4), (31, 70)]):
        ndi.uniform_filter(xd, size, mode=mode, cval=cval, output=out)
    assert np.array_equal(out.get(), full)
    with S.output_planes([(0, 0), (69, 70)]):                               # empty + one plane
        ndi.uniform_filter(xd, size, mode=mode, cval=cval, output=out)
    assert np.array_equal(out.get(), full)


def test_float64_volumes_with_rows_of_an_odd_number_of_doubles(gpu, ndi, lib):
    """float64 volumes (what nibabel's get_fdata() hands out) whose rows are an odd number of doubles: extended to whole 16-byte
    vectors (mi_extend_rows / mi_crop_rows on 8-byte elements, r5), filtered by the float64 streaming kernels, cropped -- every
    boundary mode, a fill value, origins along z / y, gaussian and uniform; against SciPy at float64 tolerance."""
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(64)
    for shape in ((45, 54, 45), (33, 40, 101), (20, 37, 263), (181, 217, 181)):
        x = rng.standard_normal(shape)
        xd = gpu.asarray(x)
        for mode in MODES:
            for what in (("u", 3), ("u", 5), ("g", 1.0), ("g", 2.0)) if shape[0] < 100 else (("u", 5), ("g", 2.0)):
                kw = dict(mode=mode, cval=-0.75)
                if what[0] == "u":
                    got = ndi.uniform_filter(xd, what[1], **kw).get()
                    ref = sndi.uniform_filter(x, what[1], **kw)
                else:
                    got = ndi.gaussian_filter(xd, what[1], **kw).get()
                    ref = sndi.gaussian_filter(x, what[1], **kw)
                assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), (shape, mode, what, float(np.abs(got - ref).max()))
        got = ndi.uniform_filter(xd, 5, origin=(1, -1, 0)).get()
        assert np.abs(got - sndi.uniform_filter(x, 5, origin=(1, -1, 0))).max() <= 1e-12 * 4
        # the route itself: the fused float64 path takes the request (it answered None for odd rows before r5)
        from cupyimg_amd.scipy.ndimage import filters as F
        if x.size >= (1 << 15):
            w = [np.full(5, 0.2)] * 3
            assert F._fused_3d_f64(xd, gpu.empty(shape, np.float64), w, [0, 0, 0], ["reflect"] * 3, 0.0) is not None, shape
    # flat min / max and grey morphology on such rows: bit-exact
    for shape in ((45, 54, 45), (20, 37, 263), (91, 109, 91)):
        x = rng.standard_normal(shape) * 40
        xd = gpu.asarray(x)
        for mode in MODES:
            for size in (3, 5, (1, 3, 5)):
                for fn, rf in ((ndi.maximum_filter, sndi.maximum_filter), (ndi.minimum_filter, sndi.minimum_filter)):
                    assert np.array_equal(fn(xd, size, mode=mode, cval=7).get(), rf(x, size, mode=mode, cval=7)), (shape, mode, size, fn.__name__)
        assert np.array_equal(ndi.grey_erosion(xd, size=3).get(), sndi.grey_erosion(x, size=3))
        assert np.array_equal(ndi.grey_dilation(xd, size=5).get(), sndi.grey_dilation(x, size=5))
    # the row copies themselves on 8-byte elements (int64 too): extend by every mode, crop back
    import ctypes
    from cupyimg_amd import core
    from cupyimg_amd.scipy.ndimage import _support as S
    for dt in (np.float64, np.int64):
        a = (rng.standard_normal((7, 9, 37)) * 1000).astype(dt)
        ad = gpu.asarray(a)
        for mode in MODES:
            ext = core.empty((7, 9, 2 + 38 + 4), dt)
            da, de = ad._desc(), ext._desc()
            S.check(lib.mi_extend_rows(ctypes.byref(da), ctypes.byref(de), 2, S.mode_code(mode), 5.0, None))
            want = np.pad(a, ((0, 0), (0, 0), (2, 5)), mode={"reflect": "symmetric", "mirror": "reflect", "nearest": "edge", "wrap": "wrap", "constant": "constant"}[mode],
                          **({"constant_values": 5} if mode == "constant" else {}))
            assert np.array_equal(ext.get(), want), (dt, mode)
            back = core.empty(a.shape, dt)
            db = back._desc()
            S.check(lib.mi_crop_rows(ctypes.byref(de), ctypes.byref(db), 2, None))
            assert np.array_equal(back.get(), a), (dt, mode)


# ---------------------------------------------------------------------------------------------------------------------
# r6: flat min / max (grey erosion / dilation, minimum / maximum filters with a cubic size) on rows that are not a multiple
# of four floats -- the same ragged kernel with min / max for its three passes (sep3d_lean_kernel<..., ragged, min | max>),
# no mi_extend_rows / mi_crop_rows around an LDS-DMA launch (filters.py:1373-1419, morphology.py:769-884)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("size", [3, 5, 7])
def test_ragged_rows_minmax_every_mode_against_scipy(gpu, ndi, lib, size):
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(600 + size)
    seen_tails = set()
    for shape in _shapes():
        if min(shape) < size and shape != (1, 64, 129):
            pass                                                    # windows longer than an axis: the boundary maps still apply
        x = rng.standard_normal(shape).astype(np.float32)
        xd = gpu.asarray(x)
        for mode in MODES:
            for fn, sfn, tag in ((ndi.minimum_filter, sndi.minimum_filter, "min"), (ndi.maximum_filter, sndi.maximum_filter, "max")):
                got = fn(xd, size=size, mode=mode, cval=0.25).get()
                k = last_kernel()
                assert "ragged,%s>" % tag in k and "sep3d_lean_kernel<%d," % size in k, (shape, mode, k)
                assert np.array_equal(got, sfn(x, size=size, mode=mode, cval=0.25)), (shape, mode, tag)
        seen_tails.add(shape[2] & 3)
        assert np.array_equal(ndi.grey_erosion(xd, size=size).get(), sndi.grey_erosion(x, size=size)), shape
        assert "ragged,min>" in last_kernel()
        assert np.array_equal(ndi.grey_dilation(xd, size=size).get(), sndi.grey_dilation(x, size=size)), shape
        assert "ragged,max>" in last_kernel()
        # origins and non-cubic sizes are not this kernel's: still SciPy's numbers through the extended-rows route
        if min(shape) >= size:
            assert np.array_equal(ndi.minimum_filter(xd, size=size, origin=(1, 0, 0)).get(), sndi.minimum_filter(x, size=size, origin=(1, 0, 0)))
    assert seen_tails == {1, 2, 3}


def test_ragged_rows_minmax_non_finite_values_and_mni_burst(gpu, ndi, lib):
    """inf / -inf / signed zeros follow SciPy; a NaN makes an output NaN exactly where the compare-select passes do (first tap of
    a pass) -- the same contract as the fused kernel on aligned rows (test_fused_float32_minmax); 181 x 217 x 181 as the last
    launch of a burst."""
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(61)
    x = rng.standard_normal((181, 217, 181)).astype(np.float32)
    xd = gpu.asarray(x)
    out = gpu.empty(x.shape, np.float32)
    for size, mode in ((3, "reflect"), (5, "mirror"), (7, "constant")):
        for _ in range(30):
            ndi.grey_erosion(xd, size=size, mode=mode, cval=-0.5, output=out)
        assert "ragged,min>" in last_kernel(), last_kernel()
        assert np.array_equal(out.get(), sndi.grey_erosion(x, size=size, mode=mode, cval=-0.5)), (size, mode)
    w = rng.standard_normal((19, 33, 183)).astype(np.float32)
    idx = rng.integers(0, w.size, size=w.size // 40)
    w.flat[idx[1::4]] = np.inf
    w.flat[idx[2::4]] = -np.inf
    w.flat[idx[3::4]] = -0.0
    wd = gpu.asarray(w)
    for fn, ref in ((ndi.minimum_filter, sndi.minimum_filter), (ndi.maximum_filter, sndi.maximum_filter)):
        assert np.array_equal(fn(wd, size=5, mode="mirror").get(), ref(w, size=5, mode="mirror")), ref.__name__
        assert "ragged" in last_kernel()
    w.flat[idx[0::4]] = np.nan
    wd = gpu.asarray(w)
    clean = sndi.maximum_filter(np.isnan(w).astype(np.uint8), size=5, mode="mirror") == 0
    for fn, ref in ((ndi.minimum_filter, sndi.minimum_filter), (ndi.maximum_filter, sndi.maximum_filter)):
        got = fn(wd, size=5, mode="mirror").get()
        assert "ragged" in last_kernel()
        lib.mi_debug_set_sep3d_ragged(0)                          # the extended-rows route: LDS-DMA kernel, same per-pass semantics
        try:
            via = fn(wd, size=5, mode="mirror").get()
        finally:
            lib.mi_debug_set_sep3d_ragged(1)
        assert np.array_equal(np.isnan(got), np.isnan(via)), ref.__name__
        assert np.array_equal(got[clean], ref(np.where(np.isnan(w), np.float32(0), w), size=5, mode="mirror")[clean]), ref.__name__


# ---------------------------------------------------------------------------------------------------------------------
# r6: uint8 cubic min / max on rows that are not a multiple of 16 bytes, as they lie (csrc/minmax3d_u8r.hip)
# ---------------------------------------------------------------------------------------------------------------------
def _u8_ragged_shapes():
    # nx % 16 = 5, 13, 15, 1, 2, 1, 3, 15, 15 (64 granules), 9 (one granule per row), 8, 14, 11, 12, 6, 7, 10, 4; ny that leaves
    # partial blocks of four rows; one and a few planes
    return [(24, 37, 181), (17, 30, 301), (9, 21, 255), (12, 19, 257), (40, 30, 66), (64, 70, 17), (50, 60, 19), (5, 20, 511),
            (3, 40, 1023), (70, 64, 9), (33, 47, 24), (11, 33, 190), (41, 43, 27), (9, 31, 172), (37, 35, 38), (1, 200, 183),
            (2, 403, 42), (47, 41, 20)]


@pytest.mark.parametrize("size", [3, 5, 7])
def test_u8_ragged_rows_minmax_every_mode_against_scipy(gpu, ndi, lib, size):
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    lib.mi_debug_set_u8_ragged.argtypes = [ctypes.c_int]
    rng = np.random.default_rng(6600 + size)
    tails = set()
    for shape in _u8_ragged_shapes():
        if shape[0] * shape[1] * shape[2] % 256 > 240:
            continue                                                  # (an array that may end within 16 bytes of its pool block: see the next test)
        x = rng.integers(0, 256, size=shape).astype(np.uint8)
        if shape[2] == 181:
            x[rng.random(shape) < 0.3] = 0                           # plateaus and both extremes
            x[rng.random(shape) < 0.1] = 255
        xd = gpu.asarray(x)
        for mode in MODES:
            for fn, sfn, tag in ((ndi.minimum_filter, sndi.minimum_filter, "min"), (ndi.maximum_filter, sndi.maximum_filter, "max")):
                got = fn(xd, size=size, mode=mode, cval=7).get()
                k = last_kernel()
                assert "mm3u8_ragged_kernel<%d,%s>" % (size, tag) in k, (shape, mode, k)
                ref = sfn(x, size=size, mode=mode, cval=7)
                assert np.array_equal(got, ref), (shape, mode, tag, int((got != ref).sum()))
        tails.add(shape[2] % 16)
        assert np.array_equal(ndi.grey_erosion(xd, size=size).get(), sndi.grey_erosion(x, size=size)), shape
        assert "mm3u8_ragged_kernel<%d,min>" % size in last_kernel()
        assert np.array_equal(ndi.grey_dilation(xd, size=size).get(), sndi.grey_dilation(x, size=size)), shape
        assert "mm3u8_ragged_kernel<%d,max>" % size in last_kernel()
        # mixed modes per axis; a user-provided output; what the kernel does not take (origins, non-cubic sizes) keeps SciPy's numbers
        modes = ("constant", "wrap", "mirror")
        out = gpu.empty(shape, np.uint8)
        ndi.minimum_filter(xd, size=size, mode=modes, cval=200, output=out)
        assert "mm3u8_ragged_kernel" in last_kernel()
        assert np.array_equal(out.get(), sndi.minimum_filter(x, size=size, mode=modes, cval=200)), shape
        assert np.array_equal(ndi.maximum_filter(xd, size=size, origin=(0, 1, 0)).get(), sndi.maximum_filter(x, size=size, origin=(0, 1, 0)))
        assert np.array_equal(ndi.minimum_filter(xd, size=(size, 3, size)).get(), sndi.minimum_filter(x, size=(size, 3, size)))
        lib.mi_debug_set_u8_ragged(0)
        try:
            via = ndi.grey_erosion(xd, size=size).get()
            assert shape[2] < 32 or "mm3u8_ragged_kernel" not in last_kernel()      # (the per-axis passes of tiny rows leave no note)
        finally:
            lib.mi_debug_set_u8_ragged(1)
        assert np.array_equal(via, sndi.grey_erosion(x, size=size)), shape
    assert len(tails) >= 12


def test_u8_ragged_rows_mni_burst_and_views(gpu, ndi, lib):
    """181 x 217 x 181 as the last launch of a burst; an array that ends exactly at the end of its allocation has no 16 readable
    bytes behind it and must take the other route with the same result; a view into a larger buffer has them."""
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(66)
    x = rng.integers(0, 256, size=(181, 217, 181)).astype(np.uint8)
    xd = gpu.asarray(x)
    out = gpu.empty(x.shape, np.uint8)
    for size, mode in ((3, "reflect"), (5, "mirror"), (7, "constant")):
        for _ in range(30):
            ndi.grey_erosion(xd, size=size, mode=mode, cval=3, output=out)
        assert "mm3u8_ragged_kernel<%d,min>" % size in last_kernel(), last_kernel()
        assert np.array_equal(out.get(), sndi.grey_erosion(x, size=size, mode=mode, cval=3)), (size, mode)
    big = gpu.asarray(rng.integers(0, 256, size=(40, 50, 77)).astype(np.uint8))
    sub = big[3:35]                                                   # contiguous, 5 planes of the buffer behind it
    got = ndi.grey_dilation(sub, size=3).get()
    assert "mm3u8_ragged_kernel" in last_kernel(), last_kernel()
    assert np.array_equal(got, sndi.grey_dilation(big.get()[3:35], size=3))
    tail = big[8:]                                                    # ends where the buffer ends (or at the pool block's end: either is fine)
    assert np.array_equal(ndi.grey_dilation(tail, size=3).get(), sndi.grey_dilation(big.get()[8:], size=3))


# ---------------------------------------------------------------------------------------------------------------------
# r6: 11 .. 17 cubic taps on rows that are not a multiple of four floats through the LDS-DMA kernel (sep3d_long3_kernel<..., ragged>)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("taps", [11, 13, 15, 17])
def test_ragged_rows_long_kernels_against_scipy_and_the_extended_rows_route(gpu, ndi, lib, taps):
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(1700 + taps)
    sigma = {11: 1.3, 13: 1.5, 15: 1.8, 17: 2.0}[taps]
    tails = set()
    # nx % 4 = 1, 2, 3; one tile and two tiles per row (the second one ragged); rows just beyond a multiple of 256; ny, nz that
    # leave partial y tiles and chunks shorter than the window
    for shape in [(24, 37, 181), (17, 30, 301), (9, 40, 253), (20, 21, 255), (12, 19, 257), (33, 18, 18), (8, 33, 19), (5, 20, 511),
                  (40, 17, 66), (19, 64, 129)]:
        x = rng.standard_normal(shape).astype(np.float32)
        xd = gpu.asarray(x)
        for mode in MODES:
            got = ndi.uniform_filter(xd, taps, mode=mode).get()
            k = last_kernel()
            assert "sep3d_long3_kernel<%d,true,ragged>" % taps in k, (shape, mode, k)
            ref = sndi.uniform_filter(x.astype(np.float64), taps, mode=mode)
            assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (shape, mode)
            got = ndi.gaussian_filter(xd, sigma, mode=mode).get()
            assert "sep3d_long3_kernel<%d,true,ragged>" % taps in last_kernel(), (shape, mode, last_kernel())
            ref = sndi.gaussian_filter(x.astype(np.float64), sigma, mode=mode)
            assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (shape, mode, "gaussian")
        tails.add(shape[2] & 3)
        # against the extended-rows route (the same kernel on rows extended to multiples of 16 bytes): same taps, same order
        lib.mi_debug_set_sep3d_ragged(0)
        try:
            via = ndi.uniform_filter(xd, taps, mode="reflect").get()
            assert x.size < (1 << 15) or "ragged" not in last_kernel()    # (small volumes: per-axis passes, which leave no note)
        finally:
            lib.mi_debug_set_sep3d_ragged(1)
        got = ndi.uniform_filter(xd, taps, mode="reflect").get()
        assert np.abs(got - via).max() <= 2e-6 * max(1.0, np.abs(via).max()), shape
        # origins along z / y; a non-zero fill value and per-axis sigmas are not this build's: still SciPy's numbers
        if min(shape[:2]) > taps:
            got = ndi.uniform_filter(xd, taps, mode="nearest", origin=(2, -3, 0)).get()
            assert "ragged" in last_kernel(), last_kernel()
            ref = sndi.uniform_filter(x.astype(np.float64), taps, mode="nearest", origin=(2, -3, 0))
            assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (shape, "origin")
        got = ndi.uniform_filter(xd, taps, mode="constant", cval=1.5).get()
        ref = sndi.uniform_filter(x.astype(np.float64), taps, mode="constant", cval=1.5)
        assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (shape, "cval")
    assert tails == {1, 2, 3}


def test_ragged_rows_long_kernel_mni_every_plane_after_a_burst(gpu, ndi, lib):
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    x = np.random.default_rng(17).standard_normal((181, 217, 181)).astype(np.float32)
    xd = gpu.asarray(x)
    out = gpu.empty(x.shape, np.float32)
    for sigma, taps in ((2.0, 17), (1.5, 13)):
        for _ in range(30):
            ndi.gaussian_filter(xd, sigma, output=out)
        assert "sep3d_long3_kernel<%d,true,ragged>" % taps in last_kernel(), last_kernel()
        ref = sndi.gaussian_filter(x.astype(np.float64), sigma)
        err = np.abs(out.get() - ref).reshape(181, -1).max(axis=1)
        assert err.max() <= 1e-6 * np.abs(ref).max(), (sigma, int(err.argmax()), float(err.max()))


@pytest.mark.parametrize("size", [3, 5, 7])
@pytest.mark.parametrize("dtype", [np.int16, np.uint16])
def test_16bit_ragged_rows_minmax_every_mode_against_scipy(gpu, ndi, lib, size, dtype):
    """uint16 / int16 volumes whose rows are not a multiple of 16 bytes (what a scanner hands out on the MNI grid), rows as they lie:
    mm3s16_ragged_kernel (csrc/minmax3d_16r.hip), every tail nx % 8, every mode, mixed modes, against SciPy bit for bit."""
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    lib.mi_debug_set_s16_ragged.argtypes = [ctypes.c_int]
    rng = np.random.default_rng(1600 + size + (7 if dtype == np.int16 else 0))
    info = np.iinfo(dtype)
    tails = set()
    for shape in [(24, 37, 181), (17, 30, 301), (9, 21, 255), (12, 19, 257), (40, 30, 66), (64, 70, 17), (50, 60, 19), (5, 20, 511),
                  (70, 64, 9), (33, 47, 28), (11, 33, 190), (9, 31, 172), (37, 35, 38), (47, 41, 20), (30, 40, 29)]:
        if shape[0] * shape[1] * shape[2] * 2 % 256 > 240:
            continue
        x = rng.integers(info.min, int(info.max) + 1, size=shape).astype(dtype)
        if shape[2] == 181:
            x[rng.random(shape) < 0.3] = info.min
            x[rng.random(shape) < 0.1] = info.max
        xd = gpu.asarray(x)
        cv = -7 if dtype == np.int16 else 40000
        for mode in MODES:
            for fn, sfn, tag in ((ndi.minimum_filter, sndi.minimum_filter, "min"), (ndi.maximum_filter, sndi.maximum_filter, "max")):
                got = fn(xd, size=size, mode=mode, cval=cv).get()
                k = last_kernel()
                assert "mm3s16_ragged_kernel<%d,%s,%s>" % (size, tag, "int16" if dtype == np.int16 else "uint16") in k, (shape, mode, k)
                ref = sfn(x, size=size, mode=mode, cval=cv)
                assert np.array_equal(got, ref), (shape, mode, tag, int((got != ref).sum()))
        tails.add(shape[2] % 8)
        assert np.array_equal(ndi.grey_erosion(xd, size=size).get(), sndi.grey_erosion(x, size=size)), shape
        assert np.array_equal(ndi.grey_dilation(xd, size=size).get(), sndi.grey_dilation(x, size=size)), shape
        assert "mm3s16_ragged_kernel" in last_kernel()
        modes = ("constant", "wrap", "mirror")
        assert np.array_equal(ndi.minimum_filter(xd, size=size, mode=modes, cval=cv).get(), sndi.minimum_filter(x, size=size, mode=modes, cval=cv)), shape
        assert np.array_equal(ndi.maximum_filter(xd, size=size, origin=(0, 1, 0)).get(), sndi.maximum_filter(x, size=size, origin=(0, 1, 0)))
        lib.mi_debug_set_s16_ragged(0)
        try:
            via = ndi.grey_erosion(xd, size=size).get()
        finally:
            lib.mi_debug_set_s16_ragged(1)
        assert np.array_equal(via, sndi.grey_erosion(x, size=size)), shape
    assert len(tails) >= 6
    if size == 3:
        x = rng.integers(info.min, int(info.max) + 1, size=(181, 217, 181)).astype(dtype)
        xd = gpu.asarray(x)
        out = gpu.empty(x.shape, dtype)
        for _ in range(20):
            ndi.grey_dilation(xd, size=3, output=out)
        assert "mm3s16_ragged_kernel" in last_kernel(), last_kernel()
        assert np.array_equal(out.get(), sndi.grey_dilation(x, size=3))


def test_ragged_rows_minmax_size_9_through_the_lds_dma_kernel(gpu, ndi, lib):
    """float32 min / max of size 9 on rows that are not a multiple of four floats: mm3f32_long_kernel<9,...,ragged> (r6, late) for the
    index-mapping modes; `constant` keeps the extended-rows route; bit-exact against SciPy either way, every tail, one and two x
    tiles, non-finite samples included."""
    import scipy.ndimage as sndi
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(909)
    tails = set()
    for shape in [(24, 37, 181), (17, 30, 301), (12, 40, 253), (20, 21, 255), (16, 19, 257), (33, 18, 18), (9, 33, 19), (9, 20, 511), (181, 217, 181)]:
        x = rng.standard_normal(shape).astype(np.float32)
        if shape[2] == 253:
            idx = rng.integers(0, x.size, size=x.size // 50)
            x.flat[idx[0::2]] = np.inf
            x.flat[idx[1::2]] = -np.inf
        xd = gpu.asarray(x)
        for mode in (MODES if shape[0] < 100 else MODES[:2]):
            for fn, sfn, tag in ((ndi.minimum_filter, sndi.minimum_filter, "min"), (ndi.maximum_filter, sndi.maximum_filter, "max")):
                got = fn(xd, size=9, mode=mode, cval=0.5).get()
                if mode != "constant":
                    assert "mm3f32_long_kernel<9,%s,ragged>" % tag in last_kernel(), (shape, mode, last_kernel())
                assert np.array_equal(got, sfn(x, size=9, mode=mode, cval=0.5)), (shape, mode, tag)
        tails.add(shape[2] & 3)
        assert np.array_equal(ndi.grey_dilation(xd, size=9).get(), sndi.grey_dilation(x, size=9)), shape
        assert "mm3f32_long_kernel<9,max,ragged>" in last_kernel()
    assert tails == {1, 2, 3}
    out = gpu.empty((181, 217, 181), np.float32)
    for _ in range(30):
        ndi.grey_erosion(xd, size=9, output=out)
    assert np.array_equal(out.get(), sndi.grey_erosion(x, size=9))
