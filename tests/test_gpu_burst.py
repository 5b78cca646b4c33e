"""Results UNDER LOAD: every kernel that counts its vector-memory operations by hand (LDS-DMA staging with `s_waitcnt vmcnt(n)`)
is launched in a burst of back-to-back calls and the output of the LAST call is compared with an independent kernel path.

Why (round 4): a wait that is one or two operations short lets a wave read LDS-DMA data a few hundred nanoseconds before it
lands.  After one launch on an idle GPU the data are always there; with the memory system loaded they sometimes are not --
`affine3d_zstream_kernel` produced a handful of wrong voxels in the first plane of a z chunk in two bursts out of three while
every single-launch test (whole-volume parity included) passed.  Comparators are the knob-selected older kernels: bit-identical
where the kernels share their arithmetic (interpolation, integer morphology), 1e-6 max-norm where the summation order differs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import os

BURST = int(os.environ.get("MI_TEST_BURST", "40"))
ROUNDS = int(os.environ.get("MI_TEST_BURST_ROUNDS", "3"))


@pytest.fixture(scope="module")
def ndi(gpu):
    from cupyimg_amd.scipy import ndimage
    return ndimage


@pytest.fixture(scope="module")
def lib(gpu):
    from cupyimg_amd import _lib
    return _lib.load()


def last_of_burst(fn, out):
    for _ in range(BURST):
        fn(out)
    return out.get()


def check(gpu, fn, shape, dtype, reference, exact, expect_kernel):
    from cupyimg_amd import last_kernel
    out = gpu.empty(shape, dtype)
    for r in range(ROUNDS):
        got = last_of_burst(fn, out)
        assert expect_kernel in last_kernel(), last_kernel()
        if exact:
            bad = np.argwhere(~((got == reference) | (np.isnan(got) & np.isnan(reference))))
            assert len(bad) == 0, (expect_kernel, r, len(bad), bad[:4].tolist())
        else:
            err = np.abs(got.astype(np.float64) - reference).max() / np.abs(reference).max()
            assert err <= 1e-6, (expect_kernel, r, err)


def test_separable_long_kernels_under_load(gpu, ndi, lib):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((192, 256, 512)).astype(np.float32)
    xd = gpu.asarray(x)
    # (r5: `constant` with a zero fill value runs on the r3 kernel -- nothing fetched beyond the array --, any other on the r2 kernel)
    for size, mode, cv, kern in ((5, "reflect", 0.0, "sep3d_long3_kernel<5"), (9, "mirror", 0.0, "sep3d_long3_kernel<9"), (17, "nearest", 0.0, "sep3d_long3_kernel<17"),
                                 (9, "constant", 0.5, "sep3d_long_kernel<9"), (13, "constant", -1.25, "sep3d_long_kernel<13"),
                                 (9, "constant", 0.0, "sep3d_long3_kernel<9"), (17, "constant", 0.0, "sep3d_long3_kernel<17")):
        lib.mi_debug_set_sep3d_long(1)                      # comparator: lean / streaming kernels
        try:
            ref = ndi.uniform_filter(xd, size, mode=mode, cval=cv).get().astype(np.float64)
        finally:
            lib.mi_debug_set_sep3d_long(0)
        check(gpu, lambda o: ndi.uniform_filter(xd, size, mode=mode, cval=cv, output=o), x.shape, np.float32, ref, False, kern)
    # several z chunks (the prologue of a chunk is where a first-step wait matters), the experimental matrix-core kernel
    for rows, zc, kern in ((0, 6, "sep3d_long3_kernel<9"), (4, 6, "sep3d_long4_kernel<9"), (4, 0, "sep3d_long4_kernel<17")):
        size = 17 if "17" in kern else 9
        lib.mi_debug_set_sep3d_long(1)
        try:
            ref = ndi.uniform_filter(xd, size).get().astype(np.float64)
        finally:
            lib.mi_debug_set_sep3d_long(0)
        lib.mi_debug_set_long_rows(rows); lib.mi_debug_set_long_zchunks(zc)
        try:
            check(gpu, lambda o: ndi.uniform_filter(xd, size, output=o), x.shape, np.float32, ref, False, kern)
        finally:
            lib.mi_debug_set_long_rows(0); lib.mi_debug_set_long_zchunks(0)


def test_minmax_kernels_under_load(gpu, ndi, lib):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((192, 256, 512)).astype(np.float32)
    xd = gpu.asarray(x)
    lib.mi_debug_set_minmax_f32_fused(0)
    try:
        ref = ndi.maximum_filter(xd, 5).get()
    finally:
        lib.mi_debug_set_minmax_f32_fused(1)
    check(gpu, lambda o: ndi.maximum_filter(xd, 5, output=o), x.shape, np.float32, ref, True, "mm3f32_long_kernel")
    del xd
    u = rng.integers(0, 256, size=(256, 256, 1024), dtype=np.uint8)
    ud = gpu.asarray(u)
    for size in (3, 7):
        lib.mi_debug_set_u8_fused(0)
        try:
            ref = ndi.grey_erosion(ud, size=size).get()
        finally:
            lib.mi_debug_set_u8_fused(1)
        check(gpu, lambda o: ndi.grey_erosion(ud, size=size, output=o), u.shape, np.uint8, ref, True, "mm3u8")


def test_interpolation_kernels_under_load(gpu, ndi, lib):
    rng = np.random.default_rng(3)
    n = 256
    x = rng.standard_normal((n, n, n)).astype(np.float32)
    xd = gpu.asarray(x)
    ang = np.deg2rad(7.0); c, s = np.cos(ang), np.sin(ang)
    ctr = (n - 1) / 2.0

    def about_centre(M):
        return M, ctr - M @ np.array([ctr] * 3) + np.array([0.5, -1.25, 2.0])

    Myx = np.diag([1.02, 1.0, 1.0]) @ np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    Mzx = np.array([[c, 0, -s], [0, 1.0, 0], [s, 0, c]])
    Mgen = np.array([[1.0, 0.03, 0.02], [-0.03, 1.0, 0.04], [0.02, -0.04, 1.0]])          # couples all three axes a little: the LDS box kernel
    a45 = np.deg2rad(50.0); c45, s45 = np.cos(a45), np.sin(a45)
    Myx45 = np.array([[1.02, 0, 0], [0, c45, -s45], [0, s45, c45]])                        # sheared window (r4b)
    Mzx45 = np.array([[c45, 0, -s45], [0, 1.0, 0], [s45, 0, c45]])
    for M, knob_off, zc, tiles, kern in ((Myx, "affine_zstream", 0, 1, "affine3d_zrect_kernel<32,0>"), (Myx, "affine_zstream", 7, 1, "affine3d_zrect_kernel<32,0>"),
                                         (Myx, "affine_zstream", 0, 64, "affine3d_z"), (Mzx, "affine_zstream", 5, 1, "affine3d_z"),
                                         (Mgen, "interp_c1", 0, 1, "affine3d_lds_kernel"),
                                         (Myx45, "affine_zstream", 0, 1, "affine3d_zstream_kernel<32,0,true>"), (Myx45, "affine_zstream", 6, 1, "affine3d_zstream_kernel<32,0,true>"),
                                         (Mzx45, "affine_zstream", 3, 1, "affine3d_zstream_kernel<32,1,true>")):
        M, off = about_centre(M)
        if knob_off == "affine_zstream":
            lib.mi_debug_set_affine_zstream(0); lib.mi_debug_set_interp_c1(5)
        else:
            lib.mi_debug_set_interp_c1(5)
        try:
            ref = ndi.affine_transform(xd, M, off, order=1, mode="constant", cval=0.25).get()
        finally:
            lib.mi_debug_set_affine_zstream(1); lib.mi_debug_set_interp_c1(1)
        lib.mi_debug_set_affine_zchunks(zc); lib.mi_debug_set_affine_zstream(tiles)
        try:
            check(gpu, lambda o: ndi.affine_transform(xd, M, off, order=1, mode="constant", cval=0.25, output=o), x.shape, np.float32, ref, True, kern)
        finally:
            lib.mi_debug_set_affine_zchunks(0); lib.mi_debug_set_affine_zstream(1)
    # map_coordinates: every instance of the z-streaming kernel, several chunkings
    M, off = about_centre(Myx)
    idx = np.indices((n, n, n), dtype=np.float32).reshape(3, -1)
    coords = (M.astype(np.float32) @ idx + off.astype(np.float32)[:, None]).reshape(3, n, n, n)
    del idx
    cd = gpu.asarray(coords)
    lib.mi_debug_set_map_zstream(0)
    try:
        ref = ndi.map_coordinates(xd, cd, order=1, mode="constant", cval=0.25).get()
    finally:
        lib.mi_debug_set_map_zstream(1)
    for knob, variant, zc, kern in ((1, 0, 0, "map_coords3d_zstream_kernel<true,8,1>"), (1, 0, 3, "map_coords3d_zstream_kernel<true,8,1>"), (1, 82, 5, "map_coords3d_zstream_kernel<true,8,2>"),
                                    (1, 41, 0, "map_coords3d_zstream_kernel<true,4,1>"), (3, 0, 5, "map_coords3d_zstream_kernel<false,8,1>")):
        lib.mi_debug_set_map_zstream(knob); lib.mi_debug_set_map_zvariant(variant); lib.mi_debug_set_map_zchunks(zc)
        try:
            check(gpu, lambda o: ndi.map_coordinates(xd, cd, order=1, mode="constant", cval=0.25, output=o), x.shape, np.float32, ref, True, kern)
        finally:
            lib.mi_debug_set_map_zstream(1); lib.mi_debug_set_map_zvariant(0); lib.mi_debug_set_map_zchunks(0)


# ---------------------------------------------------------------------------------------------------------------------
# r5: the kernels added after the first burst test (r4b) -- and one test for the whole class
# ---------------------------------------------------------------------------------------------------------------------
def test_cubic_kernels_under_load_512(gpu, ndi, lib):
    """cubic3_zfactor_kernel<0 / 1> (r5) and cubic3_zstream_kernel<0 / 1> (r4b) -- the same five-slot LDS-DMA ring, `vmcnt(2)` at
    the top of a step, late fetches -- at 7 / 30 / 80 degrees and with the BASELINE matrix's step of 1.02 planes, and
    cubic3_rowblend_kernel, on 512^3 (full tiles: the `wide` path; four z chunks per tile column): the last of BURST
    back-to-back launches against cubic3_f32_kernel (the L1-gather kernel, which counts nothing by hand) -- bit-identical for
    the r4b and row-blend kernels, 1e-6 max-norm for the factored kernel (another order of the sums: csrc/cubic_fast.hip)."""
    from helpers.burst_cases import Cases
    cases = Cases(gpu, ndi, lib, n_cubic=512)
    names = [n for n in cases.names() if n.startswith("case_cubic_")]
    assert len(names) >= 11
    for name in names:
        fn, shape, dtype, kern = getattr(cases, name)()
        ref = gpu.empty(shape, dtype)
        lib.mi_debug_set_cubic_zstream(0); lib.mi_debug_set_cubic_rowblend(0); lib.mi_debug_set_cubic_box(0)
        try:
            fn(ref)
            from cupyimg_amd import last_kernel
            assert "cubic3_f32_kernel" in last_kernel(), (name, last_kernel())
            want = ref.get()
        finally:
            lib.mi_debug_set_cubic_zstream(1); lib.mi_debug_set_cubic_rowblend(1); lib.mi_debug_set_cubic_box(1)
        del ref
        check(gpu, fn, shape, dtype, want if "zfactor" not in kern else want.astype(np.float64), "zfactor" not in kern, kern)
        del fn, want


def test_spline_prefilter_lds_rows_under_load(gpu, ndi, lib):
    """spline_filter_rows_lds_kernel (lines loaded by LDS-DMA, `vmcnt(0)` before the recursion): 512- and 536-sample lines
    (the padded modes' length), orders 2 - 5, float32 and float64 coefficients -- last of a burst bit-identical to the
    one-thread-per-line kernel (spline_filter1d_kernel: plain loads and stores, the same line code; the tiled kernel of round 2
    applies the gain at the other end and agrees to 1e-14 only).  The r5 one-sweep kernels are switched off for this test (they
    would take orders 2 / 3)."""
    rng = np.random.default_rng(11)
    lib.mi_debug_set_spline_fast(0)
    try:
        for n, dt in ((512, np.float32), (536, np.float32), (512, np.float64)):
            x = rng.standard_normal((96, 512, n)).astype(dt)
            xd = gpu.asarray(x)
            for order in (2, 3, 4, 5):
                for mode in (("mirror", "reflect") if order == 3 else ("mirror",)):
                    lib.mi_debug_set_spline_rows(0)
                    try:
                        want = ndi.spline_filter1d(xd, order, axis=-1, output=dt, mode=mode).get()
                    finally:
                        lib.mi_debug_set_spline_rows(1)
                    out = gpu.empty(x.shape, dt)
                    for r in range(ROUNDS):
                        for _ in range(BURST):
                            ndi.spline_filter1d(xd, order, axis=-1, output=out, mode=mode)
                        got = out.get()
                        bad = int(np.count_nonzero(got != want))
                        assert bad == 0, (n, dt, order, mode, r, bad)
                    del out
            del xd
    finally:
        lib.mi_debug_set_spline_fast(1)


def test_strict_wait_build_is_bit_identical_under_load(gpu):
    """ONE test for every hand-counted wait in the tree: libmi355img_strict.so is the same source with each
    `s_waitcnt vmcnt(n)`, n > 0, compiled as vmcnt(0) (csrc/common.hpp MI_VMCNT, `python -m cupyimg_amd._build --strict`).  Both
    libraries run the burst cases of tests/helpers/burst_cases.py -- sep3d_long3 / long, mm3f32_long, affine zrect / zstream
    (both stream axes), map_coordinates zstream, cubic3_zfactor<0 / 1> and cubic3_zstream<0 / 1> incl. the late-fetch step,
    row-blend -- in their own
    processes, BURST launches back to back, twice over; the SHA-256 of every last output must agree.  A count that is an
    operation short shows here whichever kernel it is in, also in kernels added later (they only have to use MI_VMCNT)."""
    import json
    import subprocess
    import sys
    import tempfile
    from cupyimg_amd import _build
    if not os.path.exists(_build.LIB_STRICT):
        _build.build(verbose=False, strict=True)
    here = os.path.dirname(os.path.abspath(__file__))
    gpu.free_all_blocks()
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for tag, libpath in (("product", None), ("strict", _build.LIB_STRICT)):
            env = dict(os.environ)
            env.pop("MI355IMG_LIB", None)
            if libpath:
                env["MI355IMG_LIB"] = libpath
            out = os.path.join(tmp, tag + ".json")
            proc = subprocess.run([sys.executable, os.path.join(here, "helpers", "burst_cases.py"), "--out", out, "--burst", str(BURST), "--rounds", "2"],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
            assert proc.returncode == 0, proc.stdout[-3000:]
            res[tag] = json.load(open(out))
    assert res["strict"]["library"].endswith("libmi355img_strict.so") and res["product"]["library"].endswith("libmi355img.so")
    assert len(res["product"]["cases"]) >= 21
    for name, c in res["product"]["cases"].items():
        s = res["strict"]["cases"][name]
        assert c["expected"] in c["kernel"], (name, c["kernel"])
        assert c["expected"] in s["kernel"], (name, s["kernel"])
        assert len(set(c["digests"])) == 1, (name, "product build: rounds differ", c["digests"])
        assert c["digests"] == s["digests"], (name, "strict and product builds differ under load")
