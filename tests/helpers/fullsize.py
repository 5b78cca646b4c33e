"""Full-size parity legs for the BASELINE.json configs (last bullet / 8d): the GPU result of a
whole 512^3 / 1024^3 / 264 x 2048^2 call is compared with scipy.ndimage on z sub-slabs (with the halo the filter
needs).  At a global edge the sub-slab edge IS the volume edge, so index-mapping boundary modes are evaluated exactly
as unsplit.  r4: the sub-slabs TILE THE WHOLE VOLUME (`whole_volume_*`): every plane of H, B, C, D, D' is compared, the
slabs spread over the host cores by a fork pool (SciPy is single-threaded; 512^3 costs about a second on the 256-core
GPU boxes, minutes on one core); `check_*_slabs` (a handful of sub-slabs) remain for the benchmark scripts.

Test infrastructure: used by tests/test_gpu_baseline_full.py and scripts/bench_configs.py (`parity` field), never by
the package.  Inputs follow SURVEY.md 8(d): N(0,1) float32 seed 0 (H, B, D), uint8 uniform seed 1 (C), the fixed
affine M = diag(1.02, 1, 1) . R_x(7 deg) about the centre plus (0.5, -1.25, 2.0) (D, D'); tolerances BASELINE.md
section 2: 1e-6 max-norm relative (filters), bit-exact (integer morphology), 2e-6 . max(1, max|ref|) (order-1
interpolation with float32 weights)."""
import numpy as np

N_H = 512
N_C = 1024
E_SLAB = (264, 2048, 2048)


def volume_f32(shape, seed=0):
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def volume_u8(shape, seed=1):
    return np.random.default_rng(seed).integers(0, 256, size=shape, dtype=np.uint8)


def slab_volume_f32(shape, seed=2, block=66):
    """E-sized float32 slab without 10 s of host RNG: one random block of planes repeated along z with a per-plane
    offset and scale (so no two planes agree and a z shift of the result would be seen)."""
    nz = shape[0]
    base = np.random.default_rng(seed).standard_normal((block,) + tuple(shape[1:]), dtype=np.float32)
    x = np.empty(shape, np.float32)
    for z0 in range(0, nz, block):
        n = min(block, nz - z0)
        k = z0 // block
        np.multiply(base[:n], np.float32(1.0 + 0.125 * k), out=x[z0:z0 + n])
        x[z0:z0 + n] += (np.float32(0.01) * np.arange(z0, z0 + n, dtype=np.float32))[:, None, None]
    return x


def affine_case(n=N_H):
    """(M, offset) of SURVEY.md 8(d)."""
    ang = np.deg2rad(7.0)
    R = np.array([[1, 0, 0], [0, np.cos(ang), -np.sin(ang)], [0, np.sin(ang), np.cos(ang)]])
    M = np.diag([1.02, 1.0, 1.0]) @ R
    ctr = (n - 1) / 2.0
    off = ctr - M @ np.array([ctr] * 3) + np.array([0.5, -1.25, 2.0])
    return M, off


def affine_coords_f32(n=N_H):
    """The same warp materialised as float32 coordinates (3, n, n, n) = 1.5 GiB, plane by plane (np.indices of the
    whole grid would need another 1.5 GiB)."""
    M, off = affine_case(n)
    Mf, of = M.astype(np.float32), off.astype(np.float32)
    coords = np.empty((3, n, n, n), np.float32)
    yy, xx = np.meshgrid(np.arange(n, dtype=np.float32), np.arange(n, dtype=np.float32), indexing="ij")
    for z in range(n):
        zf = np.float32(z)
        for a in range(3):
            # same association as (M @ idx + off): ((m0 z + m1 y) + m2 x) + off in float32
            coords[a, z] = (Mf[a, 0] * zf + Mf[a, 1] * yy + Mf[a, 2] * xx) + of[a]
    return coords


def z_slabs(nz, width=6, interior=3, extra=()):
    """[(a, b)]: the first and last `width` planes, `interior` slabs spread over the inside, plus slabs centred on
    the planes in `extra` (e.g. a 2 GiB byte-offset crossing)."""
    out = [(0, min(width, nz)), (max(nz - width, 0), nz)]
    for k in range(interior):
        c = (k + 1) * nz // (interior + 1) + (7 * k) % 5            # not aligned with any tile / chunk size
        out.append((max(c - width // 2, 0), min(c + width // 2, nz)))
    for c in extra:
        if 0 <= c < nz:
            out.append((max(c - width // 2, 0), min(c + width // 2, nz)))
    return sorted(set(out))


def ref_on_slab(x, a, b, lo, hi, fn):
    """fn(sub-slab incl. halo)[planes a..b) -- valid for index-mapping boundary modes (see module docstring)."""
    nz = x.shape[0]
    e0, e1 = max(a - lo, 0), min(b + hi, nz)
    return fn(x[e0:e1])[a - e0:a - e0 + (b - a)]


def maxnorm_rel(got, ref):
    e = np.asarray(ref, dtype=np.float64)
    d = float(np.abs(np.asarray(got, dtype=np.float64) - e).max())
    m = float(np.abs(e).max())
    return d / m if m > 0 else d


def check_filter_slabs(x, out_dev, lo, hi, fn, slabs, exact=False):
    """Worst max-norm relative error (or number of differing voxels when `exact`) of the device result `out_dev`
    against fn() on the given z sub-slabs of the host volume `x`."""
    worst = 0
    for a, b in slabs:
        ref = ref_on_slab(x, a, b, lo, hi, fn)
        got = out_dev[a:b].get()
        assert got.shape == ref.shape, (got.shape, ref.shape)
        if exact:
            worst += int(np.count_nonzero(got != ref))
        else:
            worst = max(worst, maxnorm_rel(got, ref))
    return worst


def check_map_coordinates_slabs(x, coords, out_dev, slabs):
    """max |got - ref| / max(1, max|ref|) of an order-1 `map_coordinates(mode="constant")` result on z sub-slabs of
    the OUTPUT (the whole input is gathered from)."""
    import scipy.ndimage as sndi
    worst = 0.0
    for a, b in slabs:
        ref = sndi.map_coordinates(x, coords[:, a:b], output=np.float64, order=1, mode="constant", cval=0.0,
                                   prefilter=False)
        got = out_dev[a:b].get().astype(np.float64)
        worst = max(worst, float(np.abs(got - ref).max()) / max(1.0, float(np.abs(ref).max())))
    return worst


def check_affine_slabs(x, M, off, out_dev, slabs):
    """The same for `affine_transform(order=1, mode="constant")`: output planes a..b of the full call are the
    transform with offset + M[:, 0] * a and output_shape (b - a, ny, nx)."""
    import scipy.ndimage as sndi
    worst = 0.0
    for a, b in slabs:
        ref = sndi.affine_transform(x, M, off + M[:, 0] * a, output_shape=(b - a,) + x.shape[1:], output=np.float64,
                                    order=1, mode="constant", cval=0.0, prefilter=False)
        got = out_dev[a:b].get().astype(np.float64)
        worst = max(worst, float(np.abs(got - ref).max()) / max(1.0, float(np.abs(ref).max())))
    return worst


# ---------------------------------------------------------------------------------------------------------------------
# whole-volume legs: every output plane, z sub-slabs over a fork pool
# ---------------------------------------------------------------------------------------------------------------------
_G = {}          # what the forked workers read: set before the pool is created (copy-on-write, nothing is pickled)


def _procs(procs=None):
    import os
    return max(1, min(procs or (os.cpu_count() or 1), 64))


def _jobs(nz, planes, ranges=None):
    out = []
    for a0, b0 in (ranges or [(0, nz)]):
        for a in range(a0, b0, planes):
            out.append((a, min(a + planes, b0)))
    return out


def _run_pool(worker, jobs, procs):
    """Workers are forked from this (GPU-initialised) process and only ever run NumPy / SciPy on arrays inherited
    through `_G`; they leave through os._exit (multiprocessing's fork children do), so no HIP teardown runs in them."""
    import multiprocessing as mp
    n = _procs(procs)
    if n == 1 or len(jobs) == 1:
        return [worker(j) for j in jobs]
    with mp.get_context("fork").Pool(min(n, len(jobs))) as pool:
        return pool.map(worker, jobs, chunksize=1)


def _filter_worker(job):
    a, b = job
    x, got, fn, lo, hi, exact = _G["x"], _G["got"], _G["fn"], _G["lo"], _G["hi"], _G["exact"]
    ref = ref_on_slab(x, a, b, lo, hi, fn)
    g = got[a:b]
    if exact:
        return float(np.count_nonzero(g != ref)), 0.0
    e = np.asarray(ref, dtype=np.float64)
    return float(np.abs(g.astype(np.float64) - e).max()), float(np.abs(e).max())


def whole_volume_filter(x, got, lo, hi, fn, exact=False, planes=8, procs=None, ranges=None):
    """Every plane of `got` (host array, result of the device call on `x`) against fn() on z sub-slabs of `planes`
    planes with (lo, hi) planes of context.  Returns the number of differing voxels (`exact`) or
    max|got - ref| / max|ref| over the WHOLE volume (the same norm as one comparison of the whole arrays).
    `ranges`: plane ranges to cover instead of everything (E-slab: contiguous blocks across the byte-offset crossings)."""
    assert got.shape == x.shape
    _G.update(x=x, got=got, fn=fn, lo=lo, hi=hi, exact=exact)
    try:
        res = _run_pool(_filter_worker, _jobs(x.shape[0], planes, ranges), procs)
    finally:
        _G.clear()
    if exact:
        return int(sum(r[0] for r in res))
    d, m = max(r[0] for r in res), max(r[1] for r in res)
    return d / m if m > 0 else d


def _map_worker(job):
    import scipy.ndimage as sndi
    a, b = job
    x, got, coords = _G["x"], _G["got"], _G["coords"]
    ref = sndi.map_coordinates(x, coords[:, a:b], output=np.float64, order=1, mode="constant", cval=0.0, prefilter=False)
    return float(np.abs(got[a:b].astype(np.float64) - ref).max()), float(np.abs(ref).max())


def whole_volume_map_coordinates(x, coords, got, planes=4, procs=None):
    """max |got - ref| / max(1, max|ref|) of an order-1 `map_coordinates(mode="constant")` result, every output plane."""
    _G.update(x=x, got=got, coords=coords)
    try:
        res = _run_pool(_map_worker, _jobs(got.shape[0], planes), procs)
    finally:
        _G.clear()
    return max(r[0] for r in res) / max(1.0, max(r[1] for r in res))


def _affine_worker(job):
    import scipy.ndimage as sndi
    a, b = job
    x, got, M, off = _G["x"], _G["got"], _G["M"], _G["off"]
    ref = sndi.affine_transform(x, M, off + M[:, 0] * a, output_shape=(b - a,) + x.shape[1:], output=np.float64, order=1,
                                mode="constant", cval=0.0, prefilter=False)
    return float(np.abs(got[a:b].astype(np.float64) - ref).max()), float(np.abs(ref).max())


def whole_volume_affine(x, M, off, got, planes=4, procs=None):
    """The same for `affine_transform(order=1, mode="constant")` (output planes a..b of the full call are the transform
    with offset + M[:, 0] * a and output_shape (b - a, ny, nx)), every output plane."""
    _G.update(x=x, got=got, M=M, off=off)
    try:
        res = _run_pool(_affine_worker, _jobs(got.shape[0], planes), procs)
    finally:
        _G.clear()
    return max(r[0] for r in res) / max(1.0, max(r[1] for r in res))


# ---------------------------------------------------------------------------------------------------------------------
# r5: order 3 -- SciPy's (and the reference's, interpolation.py:275,403,582,705,823) DEFAULT spline order
# ---------------------------------------------------------------------------------------------------------------------
def _affine3_worker(job):
    import scipy.ndimage as sndi
    a, b = job
    coef, got, M, off, mode, cval = _G["coef"], _G["got"], _G["M"], _G["off"], _G["mode"], _G["cval"]
    ref = sndi.affine_transform(coef, M, off + M[:, 0] * a, output_shape=(b - a,) + got.shape[1:], output=np.float64, order=3,
                                mode=mode, cval=cval, prefilter=False)
    return float(np.abs(got[a:b].astype(np.float64) - ref).max()), float(np.abs(ref).max())


def whole_volume_affine_order3(x, M, off, got, mode="constant", cval=0.0, planes=4, procs=None):
    """`affine_transform(x, M, off, order=3, mode=mode)` (prefilter included), every output plane: the B-spline coefficients
    of the WHOLE volume are SciPy's own (`spline_filter` in float64, one pass per axis, single-threaded: ~10 s for 512^3),
    the interpolation runs on z sub-slabs of the output over the fork pool.  Modes that SciPy pads before the prefilter
    (`nearest`, `grid-constant`) are not handled here."""
    import scipy.ndimage as sndi
    assert mode in ("constant", "mirror", "reflect", "wrap", "grid-wrap")
    coef = sndi.spline_filter(x.astype(np.float64), order=3, output=np.float64, mode=mode)
    _G.update(coef=coef, got=got, M=M, off=off, mode=mode, cval=cval)
    try:
        res = _run_pool(_affine3_worker, _jobs(got.shape[0], planes), procs)
    finally:
        _G.clear()
    return max(r[0] for r in res) / max(1.0, max(r[1] for r in res))


def _rotate_worker(job):
    import scipy.ndimage as sndi
    a, b = job
    x, got, angle, kw = _G["x"], _G["got"], _G["angle"], _G["kw"]
    ref = sndi.rotate(x[:, :, a:b].astype(np.float64), angle, **kw)
    g = got[:, :, a:b]
    assert g.shape == ref.shape, (g.shape, ref.shape)
    return float(np.abs(g.astype(np.float64) - ref).max()), float(np.abs(ref).max())


def whole_volume_rotate_default_axes(x, angle, got, planes=4, procs=None, **kw):
    """`rotate(x, angle, **kw)` with SciPy's default axes (1, 0): SciPy rotates every (z, y) plane of the volume on its own
    (ndimage/_interpolation.py rotate: a loop of 2-D affine transforms over the remaining axis), so the reference is exact
    on x sub-slabs -- every voxel of the output is compared."""
    assert "axes" not in kw
    _G.update(x=x, got=got, angle=angle, kw=kw)
    try:
        res = _run_pool(_rotate_worker, _jobs(x.shape[2], planes), procs)
    finally:
        _G.clear()
    return max(r[0] for r in res) / max(1.0, max(r[1] for r in res))
