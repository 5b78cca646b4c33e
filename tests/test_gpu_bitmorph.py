"""Bit-packed binary morphology with fused iterations (csrc/bitmorph3d.hip, mi_binary_erosion_fused; reference loop
morphology.py:292-322, kernel :41-128): bit-exact against the CPU oracle and scipy.ndimage -- structures, origins,
border values, masks, iteration counts that split into several fused launches, runs until stable, tile / chunk seams
(forced small tiles), rows wider than one x tile, volumes whose rows are not a multiple of 32 voxels."""
import ctypes

import numpy as np
import pytest
import scipy.ndimage as sndi

from oracle import ndimage as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ndi(gpu):
    from cupyimg_amd.scipy import ndimage
    return ndimage


@pytest.fixture()
def knob(gpu):
    from cupyimg_amd import _lib
    lib = _lib.load()
    fn = lib.mi_debug_set_bitmorph
    fn.argtypes = [ctypes.c_int] * 3
    lib.mi_debug_set_bitmorph_table.argtypes = [ctypes.c_int]
    yield fn
    fn(1, 0, 0)
    lib.mi_debug_set_bitmorph_table(0)


def _ball(r):
    g = np.indices((2 * r + 1,) * 3) - r
    return (g ** 2).sum(0) <= r * r


STRUCTS = {
    "cross": None,
    "cube3": np.ones((3, 3, 3), bool),
    "conn18": sndi.generate_binary_structure(3, 2),
    "rand537": np.random.default_rng(5).random((5, 3, 7)) > 0.4,
    "even243": np.ones((2, 4, 3), bool),
    "rand399": np.random.default_rng(6).random((3, 9, 9)) > 0.5,
    "ball2": _ball(2),
    "line_x": np.ones((1, 1, 5), bool),
    "line_z": np.ones((3, 1, 1), bool),
    "nocentre": np.array([[[0, 1, 0]], [[1, 0, 1]], [[0, 1, 0]]], bool),
}


@pytest.mark.parametrize("shape", [(20, 37, 64), (9, 50, 1040), (33, 18, 2064), (12, 21, 528), (40, 70, 96)])
@pytest.mark.parametrize("sname", list(STRUCTS))
def test_bitmorph_matches_oracle(gpu, ndi, knob, shape, sname):
    from cupyimg_amd import last_kernel
    st = STRUCTS[sname]
    import zlib
    rng = np.random.default_rng(zlib.crc32(repr((shape, sname)).encode()))
    x = rng.random(shape) > 0.3
    m = rng.random(shape) > 0.3
    xd, md = gpu.asarray(x), gpu.asarray(m)
    smin = 3 if st is None else min(st.shape)
    cases = [dict(), dict(border_value=1), dict(iterations=2), dict(iterations=3, border_value=1), dict(iterations=5),
             dict(mask=True), dict(mask=True, iterations=3), dict(mask=True, iterations=6, border_value=1),
             dict(origin=1 if smin >= 3 else 0), dict(origin=(-1, 0, 1) if smin >= 3 else 0, iterations=2)]
    from cupyimg_amd import _lib
    # the planner's tiles; 5 output rows per tile and 3 z chunks; the built-in structures once more through the run-time table
    for tiles in [(2, 0, 0), (2, 5, 3)] + ([(2, 0, 0, "table")] if sname in ("cross", "cube3", "conn18") else []):
        knob(*tiles[:3])
        _lib.load().mi_debug_set_bitmorph_table(int(len(tiles) > 3))
        want = "table" if (len(tiles) > 3 or sname not in ("cross", "cube3", "conn18")) else sname
        for fn, ofn in [(ndi.binary_erosion, orc.binary_erosion), (ndi.binary_dilation, orc.binary_dilation)]:
            for kw in cases:
                kg, ko = dict(kw), dict(kw)
                if kw.get("mask"):
                    kg["mask"], ko["mask"] = md, m
                got = fn(xd, st, **kg).get()
                assert "bitmorph3_kernel" in last_kernel(), last_kernel()
                if not (kw.get("mask") and want in ("cube3", "conn18")) and not kw.get("origin"):
                    assert "," + want + ">" in last_kernel(), (want, last_kernel())
                ref = ofn(x, st, **ko)
                assert np.array_equal(got, ref), (fn.__name__, sname, kw, tiles, int((got != ref).sum()))


def test_bitmorph_until_stable_propagation_fill_holes(gpu, ndi, knob):
    knob(2, 0, 0)
    rng = np.random.default_rng(77)
    shape = (24, 40, 96)
    u = (rng.random(shape) > 0.2).astype(np.uint8) * rng.integers(1, 255, size=shape, dtype=np.uint8)   # bytes other than 0 / 1
    assert np.array_equal(ndi.binary_erosion(gpu.asarray(u), iterations=-1).get(), orc.binary_erosion(u, iterations=-1))
    assert np.array_equal(ndi.binary_fill_holes(gpu.asarray(u)).get(), sndi.binary_fill_holes(u))
    seed = rng.random(shape) > 0.995
    mask = rng.random(shape) > 0.35
    got = ndi.binary_propagation(gpu.asarray(seed), mask=gpu.asarray(mask)).get()
    assert np.array_equal(got, sndi.binary_propagation(seed, mask=mask))
    got = ndi.binary_propagation(gpu.asarray(seed), structure=np.ones((3, 3, 3)), mask=gpu.asarray(mask), border_value=1).get()
    assert np.array_equal(got, sndi.binary_propagation(seed, structure=np.ones((3, 3, 3)), mask=mask, border_value=1))
    # opening / closing with iterations: erosion batches then dilation batches
    x = rng.random(shape) > 0.25
    for fn, sfn in [(ndi.binary_opening, sndi.binary_opening), (ndi.binary_closing, sndi.binary_closing)]:
        for it in (1, 2, 3):
            assert np.array_equal(fn(gpu.asarray(x), iterations=it).get(), sfn(x, iterations=it)), (fn.__name__, it)


@pytest.mark.parametrize("shape", [(24, 40, 96), (13, 50, 1040), (40, 33, 2064)])
def test_opening_closing_in_one_launch(gpu, ndi, knob, shape):
    """binary_opening / binary_closing (morphology.py:464-613) as ONE launch: k erosion stages, the complement, k dilation
    stages with the mirrored structure (or the other way round) -- against SciPy; structures that are not symmetric,
    masks, border values, iteration counts up to the stage limit and beyond it (two launches per half then)."""
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(shape[2])
    x = rng.random(shape) > 0.35
    m = rng.random(shape) > 0.25
    xd, md = gpu.asarray(x), gpu.asarray(m)
    structs = [None, np.ones((3, 3, 3), bool), rng.random((5, 3, 7)) > 0.4, _ball(2), np.array([[[1, 1, 0]], [[0, 1, 0]], [[0, 1, 1]]], bool)]
    for tiles in [(2, 0, 0), (2, 6, 3)]:
        knob(*tiles)
        for st in structs:
            for fn, sfn, tag in [(ndi.binary_opening, sndi.binary_opening, "opening"), (ndi.binary_closing, sndi.binary_closing, "closing")]:
                for kw in [dict(), dict(iterations=2), dict(iterations=4), dict(border_value=1), dict(mask=True, iterations=2),
                           dict(mask=True, border_value=1, iterations=3)]:
                    kg, ko = dict(kw), dict(kw)
                    if kw.get("mask"):
                        kg["mask"], ko["mask"] = md, m
                    got = fn(xd, st, **kg).get()
                    # four stages at most in one launch (more: two launches per call); large structures on x-tiled rows may not
                    # fit one tile with four stages and fall back the same way
                    if kw.get("iterations", 1) <= 2 and (st is None or st.shape == (3, 3, 3)):
                        assert "(%s)" % tag in last_kernel(), last_kernel()
                    ref = sfn(x, st, **ko)
                    assert np.array_equal(got, ref), (tag, None if st is None else st.shape, kw, tiles, int((got != ref).sum()))
    knob(2, 0, 0)
    # beyond the stage limit, even structures, origins: the two halves as separate (fused-iteration) calls
    for kw in [dict(iterations=5), dict(origin=1), dict(structure=np.ones((2, 3, 3), bool))]:
        st = kw.pop("structure", None)
        assert np.array_equal(ndi.binary_opening(xd, st, **kw).get(), sndi.binary_opening(x, st, **kw)), kw
        assert "opening" not in last_kernel()
        assert np.array_equal(ndi.binary_closing(xd, st, **kw).get(), sndi.binary_closing(x, st, **kw)), kw
    out = gpu.empty(shape, np.bool_)
    assert ndi.binary_closing(xd, iterations=2, output=out) is out
    assert np.array_equal(out.get(), sndi.binary_closing(x, iterations=2))


@pytest.mark.parametrize("shape", [(300, 1040), (77, 2064), (1000, 96), (50, 4096)])
def test_bitmorph_images(gpu, ndi, knob, shape):
    """2-D images (one-plane volumes for the same kernel): erosion / dilation with iterations, masks, origins, opening /
    closing, until-stable -- against SciPy."""
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(shape[1])
    x = rng.random(shape) > 0.3
    m = rng.random(shape) > 0.3
    xd, md = gpu.asarray(x), gpu.asarray(m)
    disk = (np.indices((5, 5)) - 2)
    disk = (disk ** 2).sum(0) <= 4
    from cupyimg_amd import _lib
    _lib.load().mi_debug_set_bitmorph_2d.argtypes = [ctypes.c_int]
    _lib.load().mi_debug_set_bitmorph_2d(2)             # every image call the kernel can take (production: fused runs only)
    for tiles in [(2, 0, 0), (2, 7, 1)]:
        knob(*tiles)
        for st in [None, np.ones((3, 3), bool), disk, np.ones((2, 3), bool), rng.random((3, 7)) > 0.4]:
            for fn, sfn in [(ndi.binary_erosion, sndi.binary_erosion), (ndi.binary_dilation, sndi.binary_dilation)]:
                for kw in [dict(), dict(iterations=3), dict(border_value=1, iterations=2), dict(mask=True, iterations=2),
                           dict(origin=(0, 1) if st is None or st.shape[1] >= 3 else 0)]:
                    kg, ko = dict(kw), dict(kw)
                    if kw.get("mask"):
                        kg["mask"], ko["mask"] = md, m
                    got = fn(xd, st, **kg).get()
                    assert "bitmorph3_kernel" in last_kernel(), last_kernel()
                    assert np.array_equal(got, sfn(x, st, **ko)), (fn.__name__, None if st is None else st.shape, kw, tiles)
            if st is None or all(n % 2 for n in st.shape):
                for fn, sfn in [(ndi.binary_opening, sndi.binary_opening), (ndi.binary_closing, sndi.binary_closing)]:
                    assert np.array_equal(fn(xd, st, iterations=2).get(), sfn(x, st, iterations=2)), (fn.__name__, tiles)
    knob(2, 0, 0)
    _lib.load().mi_debug_set_bitmorph_2d(1)
    assert np.array_equal(ndi.binary_erosion(xd, iterations=2).get(), sndi.binary_erosion(x, iterations=2))   # the production rule
    assert "bitmorph3_kernel" in last_kernel()
    # a fused batch of four, then a tail the kernel refuses for images (single iterations: byte kernel) -- found by scripts/fuzz_r6.py
    for it in (5, 6, 9):
        assert np.array_equal(ndi.binary_dilation(xd, iterations=it).get(), sndi.binary_dilation(x, iterations=it)), it
    assert np.array_equal(ndi.binary_fill_holes(xd).get(), sndi.binary_fill_holes(x))
    seed = rng.random(shape) > 0.995
    assert np.array_equal(ndi.binary_propagation(gpu.asarray(seed), mask=md).get(), sndi.binary_propagation(seed, mask=m))


@pytest.mark.parametrize("shape", [(24, 37, 181), (17, 30, 301), (40, 21, 1043), (9, 40, 70), (30, 33, 2070), (12, 19, 184)])
def test_bitmorph_ragged_rows(gpu, ndi, knob, shape):
    """Rows that are not a multiple of 16 bytes (181 x 217 x 181 masks): staged from wherever they start, the last granule's
    foreign bits replaced by the border bit, stored in 8 / 4 / 2 / 1-byte pieces -- every tail length, one and several x
    tiles; the voxels behind the array (the next allocation) must stay untouched."""
    from cupyimg_amd import last_kernel
    rng = np.random.default_rng(shape[2])
    x = rng.random(shape) > 0.3
    m = rng.random(shape) > 0.3
    xd, md = gpu.asarray(x), gpu.asarray(m)
    for tiles in [(2, 0, 0), (2, 5, 2)]:
        knob(*tiles)
        for st in [None, np.ones((3, 3, 3), bool), rng.random((3, 5, 7)) > 0.4, _ball(2)]:
            for fn, sfn in [(ndi.binary_erosion, sndi.binary_erosion), (ndi.binary_dilation, sndi.binary_dilation)]:
                for kw in [dict(), dict(border_value=1), dict(iterations=3), dict(iterations=6, border_value=1), dict(mask=True, iterations=2),
                           dict(origin=(0, 1, -1), iterations=2)]:
                    kg, ko = dict(kw), dict(kw)
                    if kw.get("mask"):
                        kg["mask"], ko["mask"] = md, m
                    if kw.get("iterations", 1) != 1:
                        ko["brute_force"] = True           # SciPy's coordinate-list path corrupts its heap on some of these
                    got = fn(xd, st, **kg).get()
                    assert "ragged> " in last_kernel(), last_kernel()
                    assert np.array_equal(got, sfn(x, st, **ko)), (fn.__name__, None if st is None else st.shape, kw, tiles)
        for fn, sfn in [(ndi.binary_opening, sndi.binary_opening), (ndi.binary_closing, sndi.binary_closing)]:
            assert np.array_equal(fn(xd, iterations=2).get(), sfn(x, iterations=2)), (fn.__name__, tiles)
            assert "ragged> " in last_kernel() and "ing)" in last_kernel(), last_kernel()
    knob(2, 0, 0)
    # output into the middle of a larger buffer: the bytes either side of it keep their pattern
    big = gpu.asarray(np.full(x.size + 64, 7, np.uint8))
    out = big[32:32 + x.size].reshape(shape)
    assert ndi.binary_dilation(xd, iterations=2, output=out) is out
    assert "ragged> " in last_kernel(), last_kernel()
    h = big.get()
    assert (h[:32] == 7).all() and (h[32 + x.size:] == 7).all()
    assert np.array_equal(h[32:32 + x.size].reshape(shape).astype(bool), sndi.binary_dilation(x, iterations=2))
    assert np.array_equal(ndi.binary_fill_holes(xd).get(), sndi.binary_fill_holes(x))
    seed = rng.random(shape) > 0.99
    assert np.array_equal(ndi.binary_propagation(gpu.asarray(seed), mask=md).get(), sndi.binary_propagation(seed, mask=m))


@pytest.mark.parametrize("shape", [(30, 44, 96), (25, 37, 181), (9, 11, 80)])
def test_cubes_and_octahedra_run_as_iterations_of_their_3x3x3_root(gpu, ndi, knob, shape):
    """ones((2r+1,)*3) = r iterations of ones((3,3,3)), the octahedron of radius r = r iterations of the cross: exact with
    either border value, also where the structure is larger than the array, and iterations multiply."""
    from cupyimg_amd import last_kernel
    knob(2, 0, 0)
    rng = np.random.default_rng(shape[2])
    for density in (0.1, 0.5, 0.9):
        x = rng.random(shape) > density
        xd = gpu.asarray(x)
        for r in (2, 3, 4):
            cube = np.ones((2 * r + 1,) * 3, bool)
            octa = np.abs(np.indices((2 * r + 1,) * 3) - r).sum(0) <= r
            for st, tag in ((cube, "cube3"), (octa, "cross")):
                for fn, sfn in [(ndi.binary_erosion, sndi.binary_erosion), (ndi.binary_dilation, sndi.binary_dilation)]:
                    for kw in (dict(), dict(border_value=1), dict(iterations=2)):
                        got = fn(xd, st, **kw).get()
                        assert "," + tag in last_kernel(), (tag, last_kernel())
                        assert np.array_equal(got, sfn(x, st, brute_force=True, **kw)), (fn.__name__, tag, r, kw, density)
                assert np.array_equal(ndi.binary_opening(xd, st).get(), sndi.binary_opening(x, st)), (tag, r)
                assert np.array_equal(ndi.binary_closing(xd, st, border_value=1).get(), sndi.binary_closing(x, st, border_value=1)), (tag, r)
        # with a mask the big structure stays what it is
        m = rng.random(shape) > 0.3
        got = ndi.binary_dilation(xd, np.ones((5, 5, 5), bool), mask=gpu.asarray(m)).get()
        assert ",table" in last_kernel(), last_kernel()
        assert np.array_equal(got, sndi.binary_dilation(x, np.ones((5, 5, 5), bool), mask=m))


@pytest.mark.parametrize("shape", [(40, 64, 96), (33, 50, 181), (70, 45, 1040), (20, 30, 2000), (64, 64, 64)])
def test_propagation_and_fill_holes_by_block_fill(gpu, ndi, knob, shape):
    """binary_propagation / binary_fill_holes (morphology.py:684-766) through bitfill3_kernel: blocks swept in place until
    stable with whole mask runs filled along x per sweep, launches repeated until no block changes -- the same fixed point
    as SciPy's one-iteration loop: random masks (tortuous paths across many blocks), smooth masks, structures with and
    without x-adjacent / diagonal taps, origins, border_value 1, ragged rows; a structure without its centre is not
    monotone and must take the iterating path."""
    from cupyimg_amd import last_kernel
    knob(2, 0, 0)
    rng = np.random.default_rng(shape[2] + shape[0])
    g = np.indices(shape).astype(np.float32)
    r2 = sum(((g[i] - (shape[i] - 1) / 2) / (0.42 * shape[i])) ** 2 for i in range(3))
    smooth = (r2 < 1.0) & (r2 > 0.4) & (rng.random(shape) > 0.02)
    assert np.array_equal(ndi.binary_fill_holes(gpu.asarray(smooth)).get(), sndi.binary_fill_holes(smooth))
    assert "bitfill3_kernel" in last_kernel(), last_kernel()
    yline = np.zeros((3, 3, 3), bool); yline[1, :, 1] = True        # no x-adjacent tap: no row fill, sweeps only
    xup = np.zeros((3, 3, 3), bool); xup[1, 1, 1] = xup[1, 1, 0] = xup[0, 1, 1] = True     # fills towards larger x only
    for density in (0.35, 0.6):
        mask = rng.random(shape) > density
        seed = (rng.random(shape) > 0.997) & mask
        sd, md = gpu.asarray(seed), gpu.asarray(mask)
        for st in [None, np.ones((3, 3, 3), bool), sndi.generate_binary_structure(3, 2), rng.random((3, 5, 3)) > 0.3, yline, xup]:
            if st is not None and not st[tuple(n // 2 for n in st.shape)]:
                st = st.copy()
                st[tuple(n // 2 for n in st.shape)] = True
            for bv in (0, 1):
                got = ndi.binary_propagation(sd, structure=st, mask=md, border_value=bv).get()
                assert "bitfill3_kernel" in last_kernel(), last_kernel()
                ref = sndi.binary_propagation(seed, structure=st, mask=mask, border_value=bv)
                assert np.array_equal(got, ref), (shape, density, None if st is None else st.shape, bv, int((got != ref).sum()))
        x = rng.random(shape) > density
        assert np.array_equal(ndi.binary_fill_holes(gpu.asarray(x)).get(), sndi.binary_fill_holes(x)), (shape, density)
        assert "bitfill3_kernel" in last_kernel(), last_kernel()
        got = ndi.binary_propagation(sd, mask=md, origin=(0, 1, 0)).get()                  # offsets that are not centred
        assert np.array_equal(got, sndi.binary_propagation(seed, mask=mask, origin=(0, 1, 0)))
        got = ndi.binary_propagation(sd, structure=np.ones((3, 3, 3)), mask=md, origin=(0, 0, 1)).get()
        assert np.array_equal(got, sndi.binary_propagation(seed, structure=np.ones((3, 3, 3)), mask=mask, origin=(0, 0, 1)))
    nocentre = np.array([[[0, 1, 0]], [[1, 0, 1]], [[0, 1, 0]]], bool)
    x = rng.random(shape) > 0.5
    m = rng.random(shape) > 0.3
    got = ndi.binary_dilation(gpu.asarray(x), nocentre, iterations=3, mask=gpu.asarray(m)).get()      # (a finite count: SciPy would oscillate forever)
    assert np.array_equal(got, sndi.binary_dilation(x, nocentre, iterations=3, mask=m, brute_force=True))


def test_bitmorph_output_forms_and_dtypes(gpu, ndi, knob):
    """int8 / uint8 inputs (any nonzero byte is true), uint8 output arrays, output given, input untouched."""
    knob(2, 0, 0)
    rng = np.random.default_rng(3)
    shape = (16, 24, 80)
    for dt in (np.uint8, np.int8, np.bool_):
        x = ((rng.random(shape) > 0.3) * rng.integers(1, 120, size=shape)).astype(dt)
        xd = gpu.asarray(x)
        out = gpu.empty(shape, np.uint8)
        assert ndi.binary_dilation(xd, iterations=3, output=out) is out
        assert np.array_equal(out.get(), sndi.binary_dilation(x, iterations=3).astype(np.uint8))
        assert np.array_equal(xd.get(), x)
        got = ndi.binary_erosion(xd, iterations=2)
        assert got.dtype == np.bool_ and np.array_equal(got.get(), sndi.binary_erosion(x, iterations=2))


def test_fused_abi_flags_and_refusals(gpu, knob):
    """mi_binary_erosion_fused directly: one flag per iteration; MI_ERR_UNSUPPORTED (nothing written) outside the envelope."""
    from cupyimg_amd import _lib
    knob(2, 0, 0)
    lib = _lib.load()
    shape = (12, 20, 64)
    x = np.zeros(shape, bool)
    x[4:8, 6:14, 20:40] = True                  # a 4 x 8 x 20 box: the cross erodes it away in 2 iterations
    xd = gpu.asarray(x)
    out = gpu.empty(shape, np.bool_)
    flags = gpu.zeros((4,), np.int32)
    st = np.ascontiguousarray(sndi.generate_binary_structure(3, 1), dtype=np.uint8)
    a, b = xd._desc(), out._desc()
    rc = lib.mi_binary_erosion_fused(ctypes.byref(a), ctypes.byref(b), st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                     (ctypes.c_int64 * 3)(3, 3, 3), (ctypes.c_int * 3)(0, 0, 0), None, 0, 0, 4,
                                     ctypes.c_void_p(flags.ptr), None)
    assert rc == 0
    assert not out.get().any()
    assert list(flags.get()) == [1, 1, 0, 0]
    # float volumes are not this kernel's
    fd = gpu.asarray(x.astype(np.float32))
    a = fd._desc()
    rc = lib.mi_binary_erosion_fused(ctypes.byref(a), ctypes.byref(b), st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                     (ctypes.c_int64 * 3)(3, 3, 3), (ctypes.c_int * 3)(0, 0, 0), None, 0, 0, 2, None, None)
    assert rc == _lib.MI_ERR_UNSUPPORTED
    a = xd._desc()
    rc = lib.mi_binary_erosion_fused(ctypes.byref(a), ctypes.byref(b), st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                     (ctypes.c_int64 * 3)(3, 3, 3), (ctypes.c_int * 3)(0, 0, 0), None, 0, 0, 0, None, None)
    assert rc == _lib.MI_ERR_INVALID_ARG


@pytest.mark.parametrize("n,iterations", [(512, 1), (512, 3), (256, 7), (1024, 2)])
def test_bitmorph_full_size_every_plane(gpu, ndi, n, iterations):
    """n^3 bool, default structure: every plane against scipy.ndimage on z sub-slabs (halo = iterations planes; at a
    global edge the slab edge is the volume edge and border_value applies as unsplit), last launch of a burst."""
    from helpers import fullsize as fs
    from cupyimg_amd import last_kernel
    gpu.free_all_blocks()
    x = np.random.default_rng(11).random((n, n, n)) > 0.2
    xd = gpu.asarray(x)
    for fn, sfn in [(ndi.binary_erosion, sndi.binary_erosion), (ndi.binary_dilation, sndi.binary_dilation)]:
        out = gpu.empty(x.shape, np.bool_)
        for _ in range(6):
            fn(xd, iterations=iterations, output=out)
        assert "bitmorph3_kernel" in last_kernel(), last_kernel()
        bad = fs.whole_volume_filter(x, out.get(), iterations, iterations, lambda s: sfn(s, iterations=iterations), exact=True,
                                     planes=16)
        assert bad == 0, (fn.__name__, bad)
    del xd, out
    gpu.free_all_blocks()
