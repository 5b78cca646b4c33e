"""Every single-GPU BASELINE.json config at FULL size through the C-ABI, checked against scipy.ndimage on EVERY plane
(r4: z sub-slabs that tile the whole volume, spread over the host cores -- tests/helpers/fullsize.py; the E-slab
included: all 264 planes, across the 2 GiB and 4 GiB byte-offset crossings).  Grid geometry (1024-wide tiles,
1.5 GiB coordinate arrays, 32-bit offset guards, > 4 GiB slabs) is only exercised at these sizes."""
import numpy as np
import pytest
import scipy.ndimage as sndi

from helpers import fullsize as fs

pytestmark = pytest.mark.gpu

# r4b: what is checked is the result of the LAST of a burst of back-to-back launches, not of one launch on an idle GPU.  A
# hand-counted `s_waitcnt vmcnt` that is one or two operations short (affine3d_zstream_kernel's first step was) reads
# LDS-DMA data a few hundred nanoseconds early: invisible after a single cold launch, a handful of wrong voxels in the
# first plane of a z chunk once the memory system is loaded -- which is how the benchmark runs the kernels.
BURST = 24


def burst(fn):
    """fn(out) -> None launched BURST times back to back into the same output (allocated by the first call)."""
    out = fn(None)
    for _ in range(BURST - 1):
        fn(out)
    return out


@pytest.fixture(scope="module")
def ndi(gpu):
    from cupyimg_amd.scipy import ndimage
    return ndimage


@pytest.fixture(scope="module")
def vol512(gpu):
    x = fs.volume_f32((fs.N_H,) * 3, seed=0)
    xd = gpu.asarray(x)
    yield x, xd
    del xd
    gpu.free_all_blocks()


def test_H_uniform5_512(gpu, ndi, vol512):
    x, xd = vol512
    from cupyimg_amd import last_kernel
    out = burst(lambda o: ndi.uniform_filter(xd, size=5, output=o))
    assert "sep3d_long3_kernel<5," in last_kernel(), last_kernel()
    out = out.get()
    err = fs.whole_volume_filter(x, out, 2, 2, lambda s: sndi.uniform_filter(s.astype(np.float64), size=5))
    assert err <= 1e-6, err


def test_H_neighbours_3_7_9_13_taps_512(gpu, ndi, vol512):
    """The tap counts next to the headline's on the same volume: 3 and 7 taps (long kernel by the r3 dispatch rule), 9 and
    13 taps, in `mirror` and `nearest` mode; chunk seams of the long kernel lie at multiples of 128 planes."""
    x, xd = vol512
    from cupyimg_amd import last_kernel
    for size, mode in ((3, "mirror"), (7, "nearest"), (9, "mirror"), (13, "reflect")):
        out = burst(lambda o: ndi.uniform_filter(xd, size=size, mode=mode, output=o))
        assert "sep3d_long3_kernel<%d," % size in last_kernel(), last_kernel()
        out = out.get()
        h = size // 2
        err = fs.whole_volume_filter(x, out, h, h, lambda s: sndi.uniform_filter(s.astype(np.float64), size=size, mode=mode))
        assert err <= 1e-6, (size, mode, err)
        del out


def test_B_gaussian_sigma2_512(gpu, ndi, vol512):
    from cupyimg_amd import last_kernel
    x, xd = vol512
    out = burst(lambda o: ndi.gaussian_filter(xd, sigma=2, output=o))
    assert "sep3d_long3_kernel<17," in last_kernel(), last_kernel()      # r6: a dispatch regression must fail, not just run slower
    out = out.get()
    # 17 taps per axis: halo 8
    err = fs.whole_volume_filter(x, out, 8, 8, lambda s: sndi.gaussian_filter(s.astype(np.float64), sigma=2), planes=16)
    assert err <= 1e-6, err


def test_D_map_coordinates_order1_512(gpu, ndi, vol512):
    x, xd = vol512
    coords = fs.affine_coords_f32(fs.N_H)
    cd = gpu.asarray(coords)
    from cupyimg_amd import last_kernel
    out = burst(lambda o: ndi.map_coordinates(xd, cd, order=1, mode="constant", output=o))
    assert "map_coords3d_zstream_kernel" in last_kernel(), last_kernel()
    assert out.dtype == np.float32 and out.shape == x.shape
    del cd
    err = fs.whole_volume_map_coordinates(x, coords, out.get())
    assert err <= 2e-6, err


def test_Dprime_affine_transform_order1_512(gpu, ndi, vol512):
    x, xd = vol512
    M, off = fs.affine_case(fs.N_H)
    from cupyimg_amd import last_kernel
    out = burst(lambda o: ndi.affine_transform(xd, M, off, order=1, mode="constant", output=o))
    assert "affine3d_zrect_kernel" in last_kernel(), last_kernel()
    err = fs.whole_volume_affine(x, M, off, out.get())
    assert err <= 2e-6, err


def test_order3_default_rotate_and_affine_512(gpu, ndi, vol512):
    """r5: order 3 is the DEFAULT of `rotate` / `affine_transform` (the reference's interpolation.py:275,403,582).  Whole-volume
    legs for the two calls the r4b kernels were written for, each the last of a burst: `rotate(vol512, 7)` with every default
    (axes (1, 0): prefilter rows_lds + two strided passes, cubic3_rowblend_kernel; reshape=True, so the output is larger than
    the input) and `affine_transform(order=3)` with the BASELINE matrix (cubic3_zfactor_kernel<0>, step 1.02 along z).
    Tolerance 2e-5 . max(1, max|ref|): float32 coefficients and weights against SciPy's double (the reference's
    `allow_float32` route, interpolation.py:330-335)."""
    from cupyimg_amd import last_kernel
    x, xd = vol512
    out = burst(lambda o: ndi.rotate(xd, 7.0, output=o))
    assert "cubic3_rowblend_kernel" in last_kernel(), last_kernel()
    assert out.dtype == np.float32
    err = fs.whole_volume_rotate_default_axes(x, 7.0, out.get())
    assert err <= 2e-5, err
    del out
    M, off = fs.affine_case(fs.N_H)
    out = burst(lambda o: ndi.affine_transform(xd, M, off, order=3, output=o))
    assert "cubic3_zfactor_kernel<0>" in last_kernel(), last_kernel()
    err = fs.whole_volume_affine_order3(x, M, off, out.get())
    assert err <= 2e-5, err


def test_C_grey_erosion7_1024_u8(gpu, ndi):
    gpu.free_all_blocks()
    u = fs.volume_u8((fs.N_C,) * 3, seed=1)
    ud = gpu.asarray(u)
    from cupyimg_amd import last_kernel
    out = burst(lambda o: ndi.grey_erosion(ud, size=7, output=o))
    assert "mm3u8_split_kernel<7,min" in last_kernel(), last_kernel()
    assert out.dtype == np.uint8
    got = out.get()
    del ud, out
    gpu.free_all_blocks()
    bad = fs.whole_volume_filter(u, got, 3, 3, lambda s: sndi.grey_erosion(s, size=7), exact=True, planes=16)
    assert bad == 0, bad


def test_E_slab_uniform9_264x2048x2048(gpu, ndi):
    """One rank's slab of config E (4.3 GiB per array): planes 128 and 256 start at 2 GiB / 4 GiB byte offsets."""
    gpu.free_all_blocks()
    x = fs.slab_volume_f32(fs.E_SLAB)
    xd = gpu.asarray(x)
    out = gpu.empty(x.shape, np.float32)
    from cupyimg_amd import last_kernel
    for _ in range(8):
        ndi.uniform_filter(xd, size=9, output=out)
    assert "sep3d_long3_kernel<9," in last_kernel(), last_kernel()
    got = out.get()
    del xd, out
    gpu.free_all_blocks()
    # every plane (the 2 GiB / 4 GiB byte-offset crossings at planes 128 / 256 and the kernel's 64-plane chunk seams included)
    err = fs.whole_volume_filter(x, got, 4, 4, lambda s: sndi.uniform_filter(s.astype(np.float64), size=9), planes=4)
    assert err <= 1e-6, err
