"""r5: the one-sweep B-spline prefilter kernels (csrc/spline_fast.hip) -- `spline_stream_kernel` (strided axes: register chunks
with a restarted anti-causal sweep) and `spline_rows_scan_kernel` (contiguous axis: prefix scans over the lanes) -- against
SciPy (the reference's own oracle for interpolation.py:105-268) and against the sequential kernels they replace.

Tolerances: float64 coefficients 1e-11 (the restart truncates at |z|^32 < 5e-19 of the data range; the sequential kernels are
held to 1e-11 / 1e-12 in tests/test_gpu_vs_oracle.py), float32 coefficients 2e-6 of the data range (rounding of the stored
type: 6e-8 per pass, three passes, gain 1.5 per pass)."""
import numpy as np
import pytest
import scipy.ndimage as sndi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ndi(gpu):
    from cupyimg_amd.scipy import ndimage
    return ndimage


@pytest.fixture(scope="module")
def lib(gpu):
    from cupyimg_amd import _lib
    return _lib.load()


def _took(fragment):
    from cupyimg_amd import last_kernel
    return fragment in last_kernel()


def test_stream_and_scan_kernels_every_line_length(gpu, ndi, lib):
    """Line lengths around every structural boundary of the two kernels: the chunk sizes (28 float32 / 12 float64), the
    look-ahead (20 / 32), n = k C + 1 (the tail chunk starts ON the last sample: its mirror end condition needs c+ of the
    chunk before), n = k C + H (the full chunk that ends the line), 256-sample scan segments, rows of 16 lanes; orders 2 and
    3, mirror and reflect ends (the modes that do not pad), strided and contiguous axes, float32 and float64 coefficients,
    float32 samples read into float64 coefficients.  Forced with the knob at 2 (the default rule wants >= 16384 lines)."""
    rng = np.random.default_rng(5)
    lib.mi_debug_set_spline_fast(2)
    lib.mi_debug_set_spline_chunk(-1)                 # few long lines are the blocked kernels' by default: not here
    try:
        # (r6: lines that are not a multiple of four samples go through the scan kernel too -- 65, 66, 67, 181, 257, 258, 1023: every tail,
        # ends in the first and in a later segment, the sample before the last in another lane / another segment)
        lengths = [64, 65, 66, 67, 68, 76, 84, 85, 96, 100, 104, 113, 124, 128, 132, 133, 140, 141, 181, 197, 252, 256, 257, 258, 260, 300, 512, 516, 1020, 1023, 1024, 2048]
        for n in lengths:
            for axis_last in (False, True):
                shape = (3, 70, n) if axis_last else (n, 5, 72)
                axis = 2 if axis_last else 0
                x = rng.standard_normal(shape)
                x[..., 0] += 3.0                      # ends that matter
                for dt, tol in ((np.float64, 1e-11), (np.float32, 2e-6)):
                    xs = x.astype(dt)
                    xd = gpu.asarray(xs)
                    for order in (2, 3):
                        for mode in ("mirror", "reflect"):
                            want = sndi.spline_filter1d(xs.astype(np.float64), order, axis=axis, mode=mode)
                            got = ndi.spline_filter1d(xd, order, axis=axis, output=dt, mode=mode)
                            assert _took("spline_rows_scan_kernel" if axis_last else "spline_stream_kernel"), (n, axis, dt)
                            err = np.abs(got.get().astype(np.float64) - want).max() / np.abs(want).max()
                            assert err <= tol, (n, axis, dt.__name__, order, mode, err)
                # float32 samples, float64 coefficients
                xs = x.astype(np.float32)
                want = sndi.spline_filter1d(xs.astype(np.float64), 3, axis=axis, mode="mirror")
                got = ndi.spline_filter1d(gpu.asarray(xs), 3, axis=axis, output=np.float64, mode="mirror").get()
                assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max(), (n, axis)
    finally:
        lib.mi_debug_set_spline_fast(1)
        lib.mi_debug_set_spline_chunk(0)


def test_volume_prefilter_takes_the_fast_kernels_and_matches_scipy(gpu, ndi, lib):
    """The default rule on volumes with many lines: spline_filter (all three axes; float32 -> float64 without the conversion
    copy; float32 -> float32), shapes whose lines are not multiples of anything, and agreement with the sequential kernels
    (knob 0) to rounding.  What the fast kernels do NOT take keeps the sequential ones: orders 4 / 5, grid-wrap ends, the
    bit-exact request behind integer outputs, lines shorter than 64 samples."""
    rng = np.random.default_rng(6)
    for shape in ((130, 132, 140), (64, 300, 68), (200, 66, 96), (97, 131, 260)):
        v = rng.standard_normal(shape).astype(np.float32)
        vd = gpu.asarray(v)
        for order in (3, 2):
            for mode in ("mirror", "reflect"):
                ref = sndi.spline_filter(v.astype(np.float64), order, mode=mode)
                got64 = ndi.spline_filter(vd, order, output=np.float64, mode=mode)
                np.testing.assert_allclose(got64.get(), ref, rtol=0, atol=1e-11 * np.abs(ref).max())
                got32 = ndi.spline_filter(vd, order, output=np.float32, mode=mode)
                np.testing.assert_allclose(got32.get(), ref, rtol=0, atol=2e-6 * np.abs(ref).max())
                lib.mi_debug_set_spline_fast(0)
                try:
                    old64 = ndi.spline_filter(vd, order, output=np.float64, mode=mode).get()
                    old32 = ndi.spline_filter(vd, order, output=np.float32, mode=mode).get()
                finally:
                    lib.mi_debug_set_spline_fast(1)
                np.testing.assert_allclose(got64.get(), old64, rtol=0, atol=1e-12 * np.abs(ref).max())
                np.testing.assert_allclose(got32.get(), old32, rtol=0, atol=1e-6 * np.abs(ref).max())
    v = rng.standard_normal((130, 132, 140)).astype(np.float32)
    vd = gpu.asarray(v)
    ndi.spline_filter(vd, 3)
    assert _took("spline_rows_scan_kernel")           # the last pass of the default call
    for order, mode in ((4, "mirror"), (5, "reflect"), (3, "grid-wrap")):
        ndi.uniform_filter(vd, 3)                      # something else in last_kernel() (the sequential passes leave no note)
        ndi.spline_filter(vd, order, mode=mode)
        assert not _took("spline_rows_scan_kernel") and not _took("spline_stream_kernel"), (order, mode)
    # the bit-exact request (behind integer outputs: SciPy's arithmetic operation for operation decides exact .5 ties) never
    # takes the restarted recursion: spline mode | 0x100 through the C-ABI
    import ctypes
    ndi.uniform_filter(vd, 3)                          # something else in last_kernel()
    c = gpu.asarray(v.astype(np.float64))
    d = c._desc()
    for axis in (0, 2):
        assert lib.mi_spline_filter1d(ctypes.byref(d), axis, 3, 0 | 0x100, None) == 0
        assert not _took("spline_rows_scan_kernel") and not _took("spline_stream_kernel"), axis
    want = v.astype(np.float64)
    for axis in (0, 2):
        want = sndi.spline_filter1d(want, 3, axis=axis, mode="mirror")
    np.testing.assert_allclose(c.get(), want, rtol=1e-12, atol=1e-12)


def test_default_order3_calls_on_a_volume_match_scipy(gpu, ndi):
    """End to end with every default (order 3, prefilter, float32 route): rotate about each axis pair, zoom, shift and a
    general affine_transform on a 160^3-class volume, against SciPy in double: 2e-5 max(1, max|ref|)."""
    rng = np.random.default_rng(8)
    v = rng.standard_normal((144, 160, 152)).astype(np.float32)
    vd = gpu.asarray(v)
    v64 = v.astype(np.float64)
    tol = lambda ref: 2e-5 * max(1.0, float(np.abs(ref).max()))
    for axes in ((1, 0), (2, 1), (0, 2)):
        for reshape in (True, False):
            got = ndi.rotate(vd, 11.0, axes=axes, reshape=reshape).get()
            ref = sndi.rotate(v64, 11.0, axes=axes, reshape=reshape)
            assert got.shape == ref.shape
            assert np.abs(got - ref).max() <= tol(ref), (axes, reshape, np.abs(got - ref).max())
    for mode in ("constant", "mirror", "reflect", "nearest"):
        got = ndi.rotate(vd, -23.0, mode=mode, reshape=False).get()
        ref = sndi.rotate(v64, -23.0, mode=mode, reshape=False)
        assert np.abs(got - ref).max() <= tol(ref), (mode, np.abs(got - ref).max())
    got = ndi.zoom(vd, (1.25, 0.8, 1.1)).get()
    ref = sndi.zoom(v64, (1.25, 0.8, 1.1))
    assert np.abs(got - ref).max() <= tol(ref)
    got = ndi.shift(vd, (2.5, -3.25, 0.75)).get()
    ref = sndi.shift(v64, (2.5, -3.25, 0.75))
    assert np.abs(got - ref).max() <= tol(ref)
    a, b = np.deg2rad(9.0), np.deg2rad(-14.0)
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    M = Rz @ Rx
    ctr = (np.array(v.shape) - 1) / 2
    off = ctr - M @ ctr
    got = ndi.affine_transform(vd, M, off).get()
    ref = sndi.affine_transform(v64, M, off)
    assert np.abs(got - ref).max() <= tol(ref)


def test_axes_that_hold_samples(gpu, ndi, lib):
    """r5: a rotation leaves the third axis to itself (unit step, integral shift): the spline is evaluated AT the samples of
    that axis, its prefilter pass and its taps cancel.  `rotate` with the default axes takes cubic3_rowblend_kernel with one
    load per row ("x holds samples"), rotations in the other two planes cubic3_zfactor_kernel with ONE plane per step;
    results against SciPy (whose own rotate filters the rotation plane only) and against the route that filters and
    interpolates every axis (interpolation._IDENT_AXES = False): float32 rounding.  Padded modes, output shapes that differ,
    shifts along the identity axis that push planes / columns outside; a matrix no single-tap kernel takes (identity axis
    but |step| > 1.3 in the plane ... ) falls back by filtering the axis after all."""
    from cupyimg_amd import last_kernel
    from cupyimg_amd.scipy.ndimage import interpolation as I
    rng = np.random.default_rng(21)
    v = rng.standard_normal((150, 140, 164)).astype(np.float32)
    vd = gpu.asarray(v)
    v64 = v.astype(np.float64)
    for axes, frag in (((1, 0), "x holds samples"), ((2, 1), "cubic3_zfactor_kernel<0>"), ((2, 0), "cubic3_zfactor_kernel<1>")):
        for mode in ("constant", "mirror", "nearest", "reflect"):
            for reshape in (False, True):
                got = ndi.rotate(vd, 17.0, axes=axes, reshape=reshape, mode=mode, cval=0.5).get()
                assert frag in last_kernel(), (axes, mode, last_kernel())
                ref = sndi.rotate(v64, 17.0, axes=axes, reshape=reshape, mode=mode, cval=0.5)
                assert got.shape == ref.shape
                assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (axes, mode, reshape, np.abs(got - ref).max())
                I._IDENT_AXES = False
                try:
                    full = ndi.rotate(vd, 17.0, axes=axes, reshape=reshape, mode=mode, cval=0.5).get()
                finally:
                    I._IDENT_AXES = True
                assert "holds samples" not in last_kernel()
                assert np.abs(got - full).max() <= 4e-6 * max(1.0, np.abs(full).max()), (axes, mode, reshape, np.abs(got - full).max())
    # integral shifts along the identity axis (planes / columns pushed outside), another output shape
    a = np.deg2rad(-9.0); c, s_ = np.cos(a), np.sin(a)
    for d, M in ((2, np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1.0]])), (0, np.array([[1.0, 0, 0], [0, c, -s_], [0, s_, c]])), (1, np.array([[c, 0, -s_], [0, 1.0, 0], [s_, 0, c]]))):
        for shift in (-7.0, 5.0):
            off = np.array([3.3, -2.1, 4.7]); off[d] = shift
            osh = (160, 150, 172)
            got = ndi.affine_transform(vd, M, off, output_shape=osh, mode="constant", cval=-1.0).get()
            assert ("holds samples" in last_kernel()) or ("cubic3_zfactor_kernel" in last_kernel()), last_kernel()
            ref = sndi.affine_transform(v64, M, off, output_shape=osh, mode="constant", cval=-1.0)
            assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (d, shift, np.abs(got - ref).max())
    # a fractional shift along the axis: no identity, the ordinary route
    M = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1.0]])
    got = ndi.affine_transform(vd, M, np.array([1.0, 2.0, 0.5])).get()
    assert "holds samples" not in last_kernel()
    ref = sndi.affine_transform(v64, M, np.array([1.0, 2.0, 0.5]))
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    # identity axis, but an in-plane scale the streaming kernels refuse (rows too wide for the rectangle): the axis is filtered after all
    M = np.array([[1.0, 0, 0], [0, 2.6 * c, -2.6 * s_], [0, 2.6 * s_, 2.6 * c]])
    got = ndi.affine_transform(vd, M, np.array([0.0, 5.0, 9.0])).get()
    ref = sndi.affine_transform(v64, M, np.array([0.0, 5.0, 9.0]))
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_prefilter_passes_under_load_512(gpu, ndi, lib):
    """Each pass of a 512^3 float32 prefilter on its own, last of 30 back-to-back launches: whole lines of the result against
    SciPy's spline_filter1d in double on sampled blocks (2e-6 of the data range), and the whole volume against the
    sequential kernel (1e-6: both round to float32).  The whole-volume SciPy leg of the composed calls is
    tests/test_gpu_baseline_full.py::test_order3_default_rotate_and_affine_512."""
    rng = np.random.default_rng(9)
    n = 512
    v = rng.standard_normal((n, n, n), dtype=np.float32)
    vd = gpu.asarray(v)
    out = gpu.empty(v.shape, np.float32)
    for axis, kern in ((0, "spline_stream_kernel"), (1, "spline_stream_kernel"), (2, "spline_rows_scan_kernel")):
        for _ in range(30):
            ndi.spline_filter1d(vd, 3, axis=axis, output=out)
        assert _took(kern), axis
        got = out.get()
        lib.mi_debug_set_spline_fast(0)
        try:
            old = ndi.spline_filter1d(vd, 3, axis=axis, output=np.float32).get()
        finally:
            lib.mi_debug_set_spline_fast(1)
        scale = float(np.abs(old).max())
        assert np.abs(got - old).max() <= 1e-6 * scale, (axis, np.abs(got - old).max() / scale)
        del old
        for lo in (0, 255, 509):
            sl = [slice(None)] * 3
            sl[(axis + 1) % 3] = slice(lo, lo + 3)
            ref = sndi.spline_filter1d(v[tuple(sl)].astype(np.float64), 3, axis=axis)
            assert np.abs(got[tuple(sl)] - ref).max() <= 2e-6 * np.abs(ref).max(), (axis, lo)


def test_zoom_shift_resampling_passes(gpu, ndi, lib):
    """r5: the separable resampling passes of diagonal order-3 transforms on volumes -- x from LDS-staged row spans
    (cubic_resample_x_lds_kernel), z with the window of planes in registers (cubic_resample_zstream_kernel) -- bit-identical to
    the r3 passes (mi_debug_set_resample_fast(0): same products, same order of the sums) for zooms in and out, anisotropic
    zooms, shifts beyond the array, every mode (array ends: reflected and cval taps take the direct path), rows that are not
    multiples of four or of the 256-output segments; and within float32 accuracy of SciPy."""
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(31)
    took = 0
    for shape in ((64, 80, 128), (50, 61, 300), (40, 44, 515), (130, 70, 72)):
        v = rng.standard_normal(shape).astype(np.float32)
        vd = gpu.asarray(v)
        calls = [("zoom", (1.3, 1.1, 1.25)), ("zoom", (0.7, 1.9, 0.55)), ("zoom", 2.0), ("shift", (0.5, -0.25, 0.75)), ("shift", (-3.2, 40.5, -70.1))]
        for name, arg in calls:
            for mode in ("constant", "mirror", "nearest", "reflect", "grid-wrap", "wrap", "grid-constant"):
                fn = getattr(ndi, name)
                got = fn(vd, arg, mode=mode, cval=0.5)
                took += "cubic_resample_zstream_kernel" in last_kernel()
                lib.mi_debug_set_resample_fast(0)
                try:
                    old = fn(vd, arg, mode=mode, cval=0.5).get()
                finally:
                    lib.mi_debug_set_resample_fast(1)
                got = got.get()
                assert np.array_equal(got, old, equal_nan=True), (shape, name, arg, mode, int(np.sum(got != old)))
                if mode in ("constant", "mirror", "nearest"):
                    ref = getattr(sndi, name)(v.astype(np.float64), arg, mode=mode, cval=0.5)
                    assert got.shape == ref.shape
                    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (shape, name, arg, mode)
    assert took >= 60, took


def test_cubic_box_kernel_general_matrices(gpu, ndi, lib):
    """r5: order-3 affine transforms whose matrix couples all three axes take their taps out of an LDS-staged box
    (cubic3_box_kernel: 16^3 output cube per workgroup) when the box fits 128 KiB (any rotation; two workgroups per CU up to 64 KiB):
    bit-identical to the gather kernel (cubic3_f32_kernel: same taps, weights, products, order of the sums) in every mode --
    voxels at the array's faces go through the kernel's second phase --, for volumes that are not multiples of the tile,
    other output shapes, padded modes, non-finite coefficients; within float32 accuracy of SciPy; matrices that shrink
    the volume by more than ~1.5 (box beyond the budget) keep the gather kernel."""
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(77)

    def rot(axis, deg):
        a = np.deg2rad(deg); u = np.asarray(axis, float); u /= np.linalg.norm(u)
        K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
        return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)

    took = 0
    for shape, oshape in (((80, 96, 112), None), ((70, 90, 132), (75, 100, 140)), ((64, 64, 256), None)):
        x = rng.standard_normal(shape).astype(np.float32)
        xd = gpu.asarray(x)
        osh = shape if oshape is None else oshape
        for axis, deg in (((1, 1, 1), 6.0), ((0.3, 1, -0.5), 9.0), ((1, 0.2, 1), -4.0), ((1, 1, 1), 38.0)):
            M = rot(axis, deg) @ np.diag([1.02, 0.98, 1.0])
            off = (np.array(shape) - 1) / 2 - M @ ((np.array(osh) - 1) / 2) + np.array([0.4, -1.3, 2.2])
            for mode in ("constant", "mirror", "nearest", "reflect", "grid-wrap", "grid-constant", "wrap"):
                for prefilter in ((True, False) if mode == "constant" else (True,)):
                    kw = dict(output_shape=osh, order=3, mode=mode, cval=0.5, prefilter=prefilter)
                    lib.mi_debug_set_cubic_box(0)
                    try:
                        want = ndi.affine_transform(xd, M, off, **kw).get()
                        assert "cubic3_f32_kernel" in last_kernel(), last_kernel()
                    finally:
                        lib.mi_debug_set_cubic_box(1)
                    got = ndi.affine_transform(xd, M, off, **kw).get()
                    took += "cubic3_box_kernel" in last_kernel()
                    assert np.array_equal(got, want, equal_nan=True), (shape, osh, axis, deg, mode, prefilter, last_kernel()[:50], int(np.sum(got != want)))
                    if prefilter and mode in ("constant", "mirror", "nearest"):
                        ref = sndi.affine_transform(x.astype(np.float64), M, off, output_shape=osh, order=3, mode=mode, cval=0.5)
                        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (shape, axis, deg, mode)
    assert took >= 80, took
    x = rng.standard_normal((80, 96, 112)).astype(np.float32)
    x[20, 30, 40] = np.inf; x[60, 70, 80] = np.nan
    xd = gpu.asarray(x)
    M = rot((1, 1, 1), 5.0); off = np.array([1.5, -2.0, 0.7])
    got = ndi.affine_transform(xd, M, off, order=3, prefilter=False).get()
    assert "cubic3_box_kernel" in last_kernel()
    ndi.affine_transform(xd, rot((1, 1, 1), 40.0) * 2.5, off, order=3, prefilter=False)
    assert "cubic3_f32_kernel" in last_kernel()
    lib.mi_debug_set_cubic_box(0)
    try:
        want = ndi.affine_transform(xd, M, off, order=3, prefilter=False).get()
    finally:
        lib.mi_debug_set_cubic_box(1)
    assert np.array_equal(got, want, equal_nan=True)


def test_cubic_mapbox_kernel(gpu, ndi, lib):
    """r5: map_coordinates with its DEFAULT order (3) on float32 volumes takes the taps out of a box every workgroup sizes from its
    own coordinates (cubic3_mapbox_kernel): bit-identical to the gather kernel (same taps, weights, products, order of the
    sums) for affine-like coordinates, smooth warps, coordinates with a bulge, +-6-voxel jitter (boxes that do not fit: the
    tile takes the gather routine), NaN / inf / 1e30 coordinates, planes wholly outside, float32 and float64 coordinates,
    every mode, prefilter on / off, shapes that are not multiples of the tile; within float32 accuracy of SciPy."""
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(123)
    took = 0
    for shape, oshape in (((64, 80, 96), (64, 80, 96)), ((50, 70, 100), (72, 66, 132))):
        x = rng.standard_normal(shape).astype(np.float32)
        xd = gpu.asarray(x)
        idx = np.indices(oshape, dtype=np.float64)
        a = np.deg2rad(8.0)
        R = np.array([[np.cos(a), -np.sin(a), 0.1], [np.sin(a), np.cos(a), -0.05], [0.02, 0.03, 0.97]])
        base = np.tensordot(R, idx, axes=1) + np.array([1.5, -3.0, 2.25])[:, None, None, None]
        fams = {"affine": base,
                "smooth": base + 2.0 * np.sin(idx[::-1] / 9.0),
                "bulge": base + 10.0 * np.exp(-((idx[0] - 30) ** 2 + (idx[1] - 30) ** 2 + (idx[2] - 40) ** 2) / 200.0)[None],
                "jitter": base + rng.uniform(-6, 6, base.shape),
                "outside": base + np.array([200.0, 0, 0])[:, None, None, None]}
        bad = base.copy()
        bad[0, 5, 6, 7] = np.nan; bad[1, 9, 9, 9] = np.inf; bad[2, 11, 3, 60] = 1e30; bad[0, 20, 21, 22] = -np.inf
        fams["nonfinite"] = bad
        for name, co in fams.items():
            for cdt in (np.float32, np.float64):
                cd = gpu.asarray(co.astype(cdt))
                for mode in (("constant", "mirror", "nearest", "reflect", "grid-wrap", "grid-constant", "wrap") if name in ("affine", "bulge") and cdt == np.float32 else ("constant", "nearest")):
                    for prefilter in ((True, False) if mode == "constant" else (True,)):
                        kw = dict(order=3, mode=mode, cval=0.5, prefilter=prefilter)
                        lib.mi_debug_set_cubic_box(0)
                        try:
                            want = ndi.map_coordinates(xd, cd, **kw).get()
                            assert "cubic3_f32_kernel" in last_kernel(), last_kernel()
                        finally:
                            lib.mi_debug_set_cubic_box(1)
                        got = ndi.map_coordinates(xd, cd, **kw).get()
                        took += "cubic3_mapbox_kernel" in last_kernel()
                        assert np.array_equal(got, want, equal_nan=True), (shape, oshape, name, cdt.__name__, mode, prefilter, int(np.sum(~((got == want) | (np.isnan(got) & np.isnan(want))))))
                        if prefilter and name in ("affine", "smooth") and mode in ("constant", "nearest"):
                            ref = sndi.map_coordinates(x.astype(np.float64), co.astype(cdt).astype(np.float64), order=3, mode=mode, cval=0.5)
                            assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (shape, name, mode)
    assert took >= 60, took
