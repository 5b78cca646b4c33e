"""Multi-GPU building blocks on ONE device: plane-restricted launches of the
fused kernel, the RCCL send/recv path (a one-rank communicator exchanging with
itself closes the chain, i.e. periodic halos) and the overlapped schedule of
SlabFilter on top of both.  The two-rank pairing itself is covered on CPU by
test_distributed_gloo.py."""
import ctypes

import numpy as np
import pytest

from _cases import maxnorm_rel
from oracle import ndimage as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ndi(gpu):
    from cupyimg_amd.scipy import ndimage
    return ndimage


@pytest.mark.parametrize("size,mode", [(3, "reflect"), (5, "reflect"), (5, "constant"), (7, "mirror"), (9, "nearest"),
                                       (9, "constant"), (13, "wrap"), (17, "reflect"),
                                       ((3, 5, 7), "wrap"), ((1, 5, 3), "reflect")])
def test_plane_restricted_launches_tile_the_full_result(gpu, ndi, size, mode):
    from cupyimg_amd.scipy.ndimage import _support as S
    rng = np.random.default_rng(11)
    x = rng.standard_normal((70, 45, 512)).astype(np.float32)
    xd = gpu.asarray(x)
    full = ndi.uniform_filter(xd, size, mode=mode, cval=0.5).get()
    sentinel = np.float32(-12345.0)
    out = gpu.asarray(np.full(x.shape, sentinel, np.float32))
    with S.output_planes([(4, 31)]):
        ndi.uniform_filter(xd, size, mode=mode, cval=0.5, output=out)
    got = out.get()
    assert np.array_equal(got[4:31], full[4:31])
    assert np.all(got[:4] == sentinel) and np.all(got[31:] == sentinel)     # nothing else written
    with S.output_planes([(0, 4), (31, 70)]):
        ndi.uniform_filter(xd, size, mode=mode, cval=0.5, output=out)
    assert np.array_equal(out.get(), full)
    with S.output_planes([(0, 0), (69, 70)]):                               # empty + one plane
        ndi.uniform_filter(xd, size, mode=mode, cval=0.5, output=out)
    assert np.array_equal(out.get(), full)


def test_plane_restriction_refuses_unfused_filters(gpu, ndi):
    from cupyimg_amd.scipy.ndimage import _support as S
    x = gpu.asarray(np.zeros((12, 12, 16), np.float64))
    with S.output_planes([(2, 4)]):
        with pytest.raises(S.Unsupported):
            ndi.uniform_filter(x, 3)
        with pytest.raises(S.Unsupported):
            ndi.minimum_filter(gpu.asarray(np.zeros((12, 12, 16), np.uint8)), 3)
    ndi.uniform_filter(x, 3)        # scope is gone


def _SelfLoopPlan(nz, lo, hi):
    """SlabPlan of a closed chain of ONE rank: both neighbours are the rank itself, so the halos are the periodic
    continuation of its own planes (distributed.SlabPlan.self_loop)."""
    from cupyimg_amd.distributed import SlabPlan
    return SlabPlan.self_loop(nz, lo, hi)


@pytest.fixture(scope="module")
def self_comm(gpu):
    from cupyimg_amd.distributed import HaloComm
    comm = HaloComm(1, 0, lambda uid: uid)
    yield comm
    comm.close()


def test_rccl_self_exchange_fills_periodic_halos(gpu, self_comm):
    rng = np.random.default_rng(12)
    nz, lo, hi = 10, 2, 3
    x = rng.standard_normal((nz, 6, 8)).astype(np.float32)
    ext = np.zeros((lo + nz + hi, 6, 8), np.float32)
    ext[lo:lo + nz] = x
    d = gpu.asarray(ext)
    self_comm.exchange(d, _SelfLoopPlan(nz, lo, hi))
    gpu.synchronize()
    want = np.concatenate([x[-lo:], x, x[:hi]])
    assert np.array_equal(d.get(), want)


@pytest.mark.parametrize("size", [3, 5, 9])
def test_overlapped_step_matches_plain_step_and_oracle(gpu, ndi, self_comm, size):
    from cupyimg_amd.distributed import SlabFilter, halo_widths
    rng = np.random.default_rng(13)
    nz = 40
    x = rng.standard_normal((nz, 33, 256)).astype(np.float32)
    lo, hi = halo_widths(size)
    plan = _SelfLoopPlan(nz, lo, hi)
    sf = SlabFilter(plan, x.shape[1:], np.float32, self_comm)
    sf.local_in[...] = gpu.asarray(x)
    fn = lambda a, b: ndi.uniform_filter(a, size=size, mode="mirror", output=b)   # noqa: E731
    plain = sf.step(fn).get()
    ref = orc.uniform_filter(x, size, mode=["wrap", "mirror", "mirror"])
    assert maxnorm_rel(plain, ref) <= 1e-6
    sf.ext_out[...] = gpu.asarray(np.zeros(sf.ext_out.shape, np.float32))
    # several steps back to back: the exchange of step k+1 must wait for the readers of step k
    for _ in range(3):
        got = sf.step_overlapped(fn, key="u")
    assert not sf._overlap_refused
    assert np.array_equal(got.get(), plain)


@pytest.mark.parametrize("nz", [40, 3])
def test_native_slab_step_matches_plain_step(gpu, ndi, self_comm, nz):
    """SlabFilter.uniform_filter / gaussian_filter: the whole overlapped step as
    one C call (mi_slab_separable3d_f32); nz = 3 is a slab without interior."""
    from cupyimg_amd.distributed import SlabFilter, halo_widths
    rng = np.random.default_rng(15)
    x = rng.standard_normal((nz, 33, 256)).astype(np.float32)
    lo, hi = halo_widths(5)
    sf = SlabFilter(_SelfLoopPlan(nz, lo, hi), x.shape[1:], np.float32, self_comm)
    sf.local_in[...] = gpu.asarray(x)
    plain = sf.step(lambda a, b: ndi.uniform_filter(a, size=5, mode="nearest", output=b)).get()
    sf.ext_out[...] = gpu.asarray(np.zeros(sf.ext_out.shape, np.float32))
    for overlap in (None, False, True):
        sf.ext_out[...] = gpu.asarray(np.zeros(sf.ext_out.shape, np.float32))
        for _ in range(3):
            got = sf.uniform_filter(5, mode="nearest", overlap=overlap)
        assert np.array_equal(got.get(), plain), overlap
    assert maxnorm_rel(plain, orc.uniform_filter(x, 5, mode=["wrap", "nearest", "nearest"])) <= 1e-6
    # gaussian sigma 0.5 -> 5 taps, same halo
    plain = sf.step(lambda a, b: ndi.gaussian_filter(a, 0.5, mode="reflect", output=b)).get()
    got = sf.gaussian_filter(0.5, mode="reflect").get()
    assert np.array_equal(got, plain)
    # 13 taps: the long fused kernel, with plane ranges as well
    if nz >= 6:
        sf2 = SlabFilter(_SelfLoopPlan(nz, 6, 6), x.shape[1:], np.float32, self_comm)
        sf2.local_in[...] = gpu.asarray(x)
        for overlap in (None, True):      # plain / overlapped native schedule
            got = sf2.uniform_filter(13, mode="mirror", overlap=overlap).get()
            assert maxnorm_rel(got, orc.uniform_filter(x, 13, mode=["wrap", "mirror", "mirror"])) <= 1e-6


def test_slab_schedule_is_measured_in_the_first_call(gpu, ndi, self_comm):
    """overlap=None: the first call of a filter (`warm`) probes the plain and the overlapped schedule (one warm step
    and three timed steps each, medians decide); later steps run the chosen one with no host synchronisation; every
    step gives the same (bit-identical) planes."""
    from cupyimg_amd.distributed import SlabFilter, halo_widths
    rng = np.random.default_rng(21)
    nz = 40
    x = rng.standard_normal((nz, 48, 256)).astype(np.float32)
    lo, hi = halo_widths(5)
    sf = SlabFilter(_SelfLoopPlan(nz, lo, hi), x.shape[1:], np.float32, self_comm)
    sf.local_in[...] = gpu.asarray(x)
    want = sf.uniform_filter(5, mode="nearest", overlap=False).get()
    sf.warm(lambda: sf.uniform_filter(5, mode="nearest"))
    (st,) = sf._tuning.values()
    assert st["choice"] in (0, 1) and st["overlap_supported"]
    assert all(len(t) == 3 and min(t) > 0 for t in st["t"]) and len(st["median_ms"]) == 2
    assert sf.schedule_of("uniform")["choice"] == st["choice"]
    outs = [sf.uniform_filter(5, mode="nearest").get() for _ in range(4)]
    assert all(np.array_equal(o, want) for o in outs)
    sf.autotune = False
    assert np.array_equal(sf.uniform_filter(5, mode="nearest").get(), want)


def test_slab_schedule_when_the_overlapped_form_is_refused(gpu, ndi, self_comm):
    """Kernels the plane-range launches do not take (19 .. 33 taps): the tuning pins the plain schedule instead of
    alternating between a refused probe and the fallback for ever (round-2 advisor finding), and every step still
    performs exactly one exchange."""
    from cupyimg_amd.distributed import SlabFilter, halo_widths
    rng = np.random.default_rng(22)
    nz = 48
    x = rng.standard_normal((nz, 40, 256)).astype(np.float32)
    lo, hi = halo_widths(21)
    sf = SlabFilter(_SelfLoopPlan(nz, lo, hi), x.shape[1:], np.float32, self_comm)
    sf.local_in[...] = gpu.asarray(x)
    ref = orc.uniform_filter(x, 21, mode=["wrap", "reflect", "reflect"])
    for _ in range(3):
        got = sf.uniform_filter(21).get()
        assert maxnorm_rel(got, ref) <= 1e-6
    st = sf.schedule_of("uniform")
    if st is not None:                       # native plain schedule taken: the overlapped probe was refused once
        assert st["choice"] == 0 and not st["overlap_supported"]
    else:                                    # not even the plain native step: remembered, generic step from then on
        assert len(sf._native_refused) == 1


def test_refusals_come_before_anything_is_queued_and_do_not_depend_on_the_rank(gpu, ndi, self_comm):
    """r3 advisor finding: with a 25-tap gaussian a rank WITHOUT interior planes (n_local <= lo + hi) used to queue its
    exchange and only then learn from the edge launch that the kernel takes no plane ranges, while ranks with an
    interior refused up front -- unequal exchange counts.  Now the kernels are asked first (dry run), with the same
    answer whatever the plane ranges: mi_separable3d_f32_supports, and the overlapped native step of a thin and of a
    thick slab both return Unsupported with an unchanged output slab."""
    import ctypes
    from cupyimg_amd import _lib
    from cupyimg_amd.distributed import SlabFilter, SlabPlan, halo_widths
    from cupyimg_amd.scipy.ndimage.filters import _gaussian_weights
    w = _gaussian_weights(3.0, 0, 4.0)
    assert len(w) == 25
    lo, hi = halo_widths(25)
    rng = np.random.default_rng(23)
    dp = ctypes.POINTER(ctypes.c_double)
    ptrs = (dp * 3)(*[w.ctypes.data_as(dp)] * 3)
    ints = lambda v: (ctypes.c_int * 3)(*v)           # noqa: E731
    lib = _lib.load()
    for nz in (lo + hi, 3 * (lo + hi)):               # no interior planes / plenty
        x = rng.standard_normal((nz, 24, 256)).astype(np.float32)
        sf = SlabFilter(SlabPlan.self_loop(nz, lo, hi), x.shape[1:], np.float32, self_comm)
        sf.local_in[...] = gpu.asarray(x)
        a, b = sf.ext_in._desc(), sf.ext_out._desc()
        args = (ctypes.byref(a), ctypes.byref(b), ptrs, ints([25] * 3), ints([0] * 3), ints([0] * 3), 0.0)
        assert lib.mi_separable3d_f32_supports(*args, 1) == _lib.MI_ERR_UNSUPPORTED        # no plane ranges at 25 taps
        assert lib.mi_separable3d_f32_supports(*args, 0) == _lib.MI_OK                     # the streaming passes take it
        sf.ext_out.fill(-7.0)
        sf._streams()
        rc = lib.mi_slab_separable3d_f32(self_comm._comm, *args, lo, hi, 0, 0, 1, sf._comm_stream.handle, sf._input_free._e,
                                         sf._halos_ready._e, None)
        assert rc == _lib.MI_ERR_UNSUPPORTED
        gpu.synchronize()
        assert np.all(sf.ext_out.get() == -7.0)                                            # nothing ran
        ref = orc.gaussian_filter(x, 3.0, mode=["wrap", "reflect", "reflect"])
        for overlap in (True, None, False):           # overlap=True: refused up front, plain schedule in its place
            got = sf.gaussian_filter(3.0, overlap=overlap).get()
            assert maxnorm_rel(got, ref) <= 1e-6, (nz, overlap)
        st = sf.schedule_of("gaussian")
        assert st is not None and st["choice"] == 0 and not st["overlap_supported"]
        assert not sf._native_refused and len(sf._overlap_refused) == 1      # native plain step, not the generic one


@pytest.mark.parametrize("nbuf", [2, 3])
@pytest.mark.parametrize("size,mode", [(5, "reflect"), (9, "nearest"), (21, "mirror")])
def test_pipelined_schedule_is_bit_identical_to_the_plain_one(gpu, ndi, self_comm, nbuf, size, mode):
    """SlabPipeline (mi_slab_pipe_*): a SEQUENCE of different volumes streams through `nbuf` resident input slabs, the
    exchange of volume j + nbuf - 1 queued before the filter of volume j; every result equals the plain step of the same
    volume bit for bit and the oracle with `wrap` along z (self-loop chain).  21 taps: the kernel takes no plane ranges
    (streaming passes on the whole extended slab)."""
    from cupyimg_amd.distributed import SlabFilter, SlabPlan, halo_widths
    rng = np.random.default_rng(50 + size)
    nz = 40
    lo, hi = halo_widths(size)
    sf = SlabFilter(SlabPlan.self_loop(nz, lo, hi), (24, 256), np.float32, self_comm)
    pipe = sf.uniform_pipeline(size, mode=mode, nbuf=nbuf)
    assert pipe.info()["planes_ok"] == (size <= 17)
    vols = [rng.standard_normal((nz, 24, 256)).astype(np.float32) for _ in range(7)]
    want = []
    for v in vols:
        sf.local_in[...] = gpu.asarray(v)
        want.append(sf.uniform_filter(size, mode=mode, overlap=False).get())
    assert maxnorm_rel(want[0], orc.uniform_filter(vols[0], size, mode=["wrap", mode, mode])) <= 1e-6
    depth = nbuf - 1
    got = []
    for j in range(len(vols) + depth):
        if j < len(vols):
            pipe.local_in(j % nbuf)[...] = gpu.asarray(vols[j])
            pipe.submit(j % nbuf)
        if j >= depth:
            got.append(pipe.compute((j - depth) % nbuf).get())
    assert len(got) == len(vols)
    for j, (g, w_) in enumerate(zip(got, want)):
        assert np.array_equal(g, w_), j
    with pytest.raises(RuntimeError):
        pipe.compute(0)                    # never submitted again
    pipe.close()


@pytest.mark.parametrize("graph", [0, 1, 6])
def test_pipelined_rotation_over_resident_inputs(gpu, ndi, self_comm, graph):
    """mi_slab_pipe_run: the rotation bench.py times (resident inputs, submits one step ahead), queued directly or
    replayed from a captured hipGraph of one / three rotations; odd step counts, repeated calls; the output is that
    of the input filtered last."""
    from cupyimg_amd.distributed import SlabFilter, SlabPlan, halo_widths
    rng = np.random.default_rng(60)
    nz, size = 36, 5
    lo, hi = halo_widths(size)
    sf = SlabFilter(SlabPlan.self_loop(nz, lo, hi), (32, 256), np.float32, self_comm)
    pipe = sf.uniform_pipeline(size, nbuf=2)
    vols = [rng.standard_normal((nz, 32, 256)).astype(np.float32) for _ in range(2)]
    want = []
    for k, v in enumerate(vols):
        sf.local_in[...] = gpu.asarray(v)
        want.append(sf.uniform_filter(size, overlap=False).get())
    for k, v in enumerate(vols):
        pipe.local_in(k)[...] = gpu.asarray(v)
    total = 0
    for n in (1, 2, 7, 12, 13, 1):
        pipe.run(n, graph)
        total += n
        assert np.array_equal(pipe.local_out.get(), want[(total - 1) % 2]), (n, total)
    info = pipe.info()
    assert info["graph_state"] in ((0,) if graph == 0 else (1, -1))       # -1: capture refused by this RCCL: direct queuing took over
    if info["graph_state"] == 1:
        assert info["graph_steps"] == max(2, graph // 2 * 2)
    k_us, ex_us = pipe.measure(reps=5)
    assert k_us > 0 and ex_us > 0
    pipe.close()


def test_slab_step_refuses_kernels_wider_than_the_halo(gpu, ndi, self_comm):
    """A plan built for 5 taps refuses a 17-tap axis-0 kernel in Python (ValueError) and in C
    (MI_ERR_INVALID_ARG), instead of filtering across the slab edge."""
    import ctypes
    from cupyimg_amd import _lib
    from cupyimg_amd.distributed import SlabFilter, halo_widths
    lo, hi = halo_widths(5)
    sf = SlabFilter(_SelfLoopPlan(40, lo, hi), (33, 256), np.float32, self_comm)
    sf.local_in[...] = gpu.asarray(np.ones((40, 33, 256), np.float32))
    with pytest.raises(ValueError):
        sf.gaussian_filter(2.0)                       # 17 taps
    with pytest.raises(ValueError):
        sf.separable([np.full(5, 0.2), None, None], origins=(1, 0, 0))
    sf.uniform_filter((5, 17, 17))                    # long kernels along y / x are fine
    # the C entry point checks on its own
    w = np.full(17, 1.0 / 17)
    dp = ctypes.POINTER(ctypes.c_double)
    ptrs = (dp * 3)(w.ctypes.data_as(dp), ctypes.cast(None, dp), ctypes.cast(None, dp))
    ints = lambda v: (ctypes.c_int * 3)(*v)           # noqa: E731
    a, b = sf.ext_in._desc(), sf.ext_out._desc()
    rc = _lib.load().mi_slab_separable3d_f32(self_comm._comm, ctypes.byref(a), ctypes.byref(b), ptrs, ints([17, 0, 0]),
                                            ints([0, 0, 0]), ints([0, 0, 0]), 0.0, lo, hi, 0, 0, 0, None, None, None, None)
    assert rc == _lib.MI_ERR_INVALID_ARG and "halo" in _lib.last_error()


def test_overlapped_step_with_long_kernels(gpu, ndi, self_comm):
    """13 taps: the fused long kernel (sep3d_long.hip) takes plane ranges, so the overlapped schedule runs and is
    bit-identical to the plain one (also in `constant` mode, which the kernel handles by zero fill + a coverage
    correction); a NON-cubic long kernel is served by the streaming passes, they take no plane ranges and the step
    falls back to the plain schedule."""
    from cupyimg_amd.distributed import SlabFilter, halo_widths
    rng = np.random.default_rng(14)
    nz, size = 40, 13
    x = rng.standard_normal((nz, 20, 256)).astype(np.float32)
    lo, hi = halo_widths(size)
    sf = SlabFilter(_SelfLoopPlan(nz, lo, hi), x.shape[1:], np.float32, self_comm)
    sf.local_in[...] = gpu.asarray(x)
    fn = lambda a, b: ndi.uniform_filter(a, size=size, mode="nearest", output=b)   # noqa: E731
    plain = sf.step(fn).get()
    got = sf.step_overlapped(fn, key="fn").get()
    assert not sf._overlap_refused
    assert np.array_equal(got, plain)
    ref = orc.uniform_filter(x, size, mode=["wrap", "nearest", "nearest"])
    assert maxnorm_rel(got, ref) <= 1e-6
    fc = lambda a, b: ndi.uniform_filter(a, size=size, mode=["wrap", "constant", "constant"], cval=0.5, output=b)   # noqa: E731
    got = sf.step_overlapped(fc, key="fc").get()
    assert not sf._overlap_refused and np.array_equal(got, sf.step(fc).get())
    ref = orc.uniform_filter(x, size, mode=["wrap", "constant", "constant"], cval=0.5)
    assert maxnorm_rel(got, ref) <= 1e-6
    fm = lambda a, b: ndi.uniform_filter(a, size=(size, 5, 5), mode="nearest", output=b)   # noqa: E731
    got = sf.step_overlapped(fm, key="fm").get()
    assert sf._overlap_refused == {"fm"}                       # remembered for THIS filter only (r3 advisor finding)
    ref = orc.uniform_filter(x, (size, 5, 5), mode=["wrap", "nearest", "nearest"])
    assert maxnorm_rel(got, ref) <= 1e-6
    assert np.array_equal(sf.step_overlapped(fn, key="fn").get(), plain) and sf._overlap_refused == {"fm"}


def test_arrays_differ(gpu):
    a = gpu.asarray(np.arange(24, dtype=np.float32).reshape(2, 3, 4))
    b = a.copy()
    assert not gpu.arrays_differ(a, b)
    b[1, 2, 3:4] = gpu.asarray(np.array([-1.0], np.float32))
    assert gpu.arrays_differ(a, b)
    assert not gpu.arrays_differ(a[0:1], b[0:1])


def test_slab_minmax_and_binary_morphology(gpu, ndi, self_comm):
    """SlabFilter.grey_erosion / maximum_filter / binary_erosion / binary_dilation on a closed one-rank chain (the halos
    are the periodic continuation): uint8 min / max bit-exact against the oracle with `wrap` on axis 0; iterated
    binary erosion with ONE exchange of iterations x reach planes; until-stable with the OR callback."""
    from cupyimg_amd.distributed import SlabFilter
    rng = np.random.default_rng(30)
    nz = 24
    u = rng.integers(0, 256, size=(nz, 40, 64)).astype(np.uint8)
    sf = SlabFilter(_SelfLoopPlan(nz, 3, 3), u.shape[1:], np.uint8, self_comm)
    sf.local_in[...] = gpu.asarray(u)
    got = sf.grey_erosion(7, mode="reflect").get()
    assert np.array_equal(got, orc.grey_erosion(u, size=7, mode=["wrap", "reflect", "reflect"]))
    got = sf.maximum_filter((5, 3, 7), mode="nearest").get()
    assert np.array_equal(got, orc.maximum_filter(u, size=(5, 3, 7), mode=["wrap", "nearest", "nearest"]))
    with pytest.raises(ValueError):
        sf.minimum_filter(9)                                   # reach 4 > halo 3

    b = rng.random((nz, 30, 64)) > 0.3
    st = orc.generate_binary_structure(3, 1)
    sfb = SlabFilter(_SelfLoopPlan(nz, 2, 2), b.shape[1:], np.bool_, self_comm)
    sfb.local_in[...] = gpu.asarray(b)
    # reference: the periodic volume = three copies stacked, middle one compared
    b3 = np.concatenate([b, b, b])
    for it in (1, 2):
        ref = orc.binary_erosion(b3, structure=st, iterations=it)[nz:2 * nz]
        assert np.array_equal(sfb.binary_erosion(st, iterations=it).get(), ref), it
        ref = orc.binary_dilation(b3, structure=st, iterations=it)[nz:2 * nz]
        assert np.array_equal(sfb.binary_dilation(st, iterations=it).get(), ref), it
    with pytest.raises(ValueError):
        sfb.binary_erosion(st, iterations=3)                   # 3 x 1 planes > halo 2
    calls = []
    res = sfb.binary_erosion(st, iterations=0, border_value=1, any_changed=lambda f: (calls.append(f), f)[1]).get()
    n_it = len(calls)
    ref = b3
    for _ in range(n_it):
        ref = orc.binary_erosion(ref, structure=st, iterations=1, border_value=1)
    # periodic in z, border_value=1 in y / x: iterate the stacked volume as often (what the ends of the stack get wrong
    # moves one plane per iteration: the middle copy is exact for n_it <= nz iterations)
    assert n_it >= 2 and calls[-1] is False and n_it <= nz
    assert np.array_equal(res, ref[nz:2 * nz])


def test_minmax_plane_ranges_and_overlapped_slab_step(gpu, ndi, self_comm):
    """mi_minmax3d_u8_planes / mi_minmax3d_f32_planes: output planes produced range by range equal the whole-volume
    launch bit for bit (uint8 cubic 3 / 5 / 7, float32 cubic 3 .. 9, every index-mapping mode, ranges that cut chunks);
    what the fused kernels do not take raises Unsupported; SlabFilter's overlapped min / max step equals the plain one."""
    from cupyimg_amd.scipy.ndimage import _support as S
    from cupyimg_amd.distributed import SlabFilter
    rng = np.random.default_rng(40)
    u = rng.integers(0, 256, size=(37, 40, 128)).astype(np.uint8)
    f = rng.standard_normal((37, 33, 264)).astype(np.float32)
    for x, sizes in ((u, (3, 5, 7)), (f, (3, 5, 7, 9))):
        xd = gpu.asarray(x)
        for size in sizes:
            for mode in ("reflect", "nearest", "mirror", "wrap"):
                for fn in (ndi.minimum_filter, ndi.maximum_filter):
                    full = fn(xd, size=size, mode=mode).get()
                    out = gpu.zeros(x.shape, x.dtype)
                    for ranges in ([(0, 5), (5, 30)], [(30, 37)]):
                        with S.output_planes(ranges):
                            fn(xd, size=size, mode=mode, output=out)
                    assert np.array_equal(out.get(), full), (x.dtype, size, mode, fn.__name__)
    with pytest.raises(S.Unsupported):
        with S.output_planes([(0, 4)]):
            ndi.minimum_filter(gpu.asarray(u), size=(3, 5, 5))          # not cubic: no plane-restricted kernel
    with pytest.raises(S.Unsupported):
        with S.output_planes([(0, 4)]):
            ndi.minimum_filter(gpu.asarray(u.astype(np.int16)), size=3)
    nz = u.shape[0]
    sf = SlabFilter(_SelfLoopPlan(nz, 3, 3), u.shape[1:], np.uint8, self_comm)
    sf.local_in[...] = gpu.asarray(u)
    plain = sf.grey_erosion(7).get()
    assert np.array_equal(sf.grey_erosion(7, overlap=True).get(), plain)
    assert np.array_equal(plain, orc.grey_erosion(u, size=7, mode=["wrap", "reflect", "reflect"]))
    sff = SlabFilter(_SelfLoopPlan(nz, 2, 2), f.shape[1:], np.float32, self_comm)
    sff.local_in[...] = gpu.asarray(f)
    plain = sff.maximum_filter(5, mode="mirror").get()
    assert np.array_equal(sff.maximum_filter(5, mode="mirror", overlap=True).get(), plain)
    assert np.array_equal(sff.minimum_filter((5, 3, 3), overlap=True).get(), sff.minimum_filter((5, 3, 3)).get())   # falls back


def test_slab_dense_correlate_and_footprint_filters(gpu, ndi, self_comm):
    """SlabFilter.correlate / convolve / minimum_filter(footprint=) / grey_erosion(structure=) / grey_dilation(footprint=):
    the halo follows from the window's axis-0 extent and origin (mirrored for convolutions and dilations), checked
    against the plan; results on a closed one-rank chain equal the oracle with `wrap` along axis 0."""
    from cupyimg_amd.distributed import SlabFilter, SlabPlan, halo_widths
    rng = np.random.default_rng(70)
    nz = 30
    x = rng.standard_normal((nz, 20, 64)).astype(np.float32)
    w = rng.standard_normal((4, 3, 5))
    for conv, origin in [(False, (0, 0, 0)), (False, (1, 0, -1)), (True, (0, 0, 0)), (True, (-2, 1, 0))]:
        o0 = origin[0] if not conv else -origin[0] - 1           # 4 taps along axis 0: even
        lo, hi = halo_widths(4, o0)
        sf = SlabFilter(SlabPlan.self_loop(nz, lo, hi), x.shape[1:], np.float32, self_comm)
        sf.local_in[...] = gpu.asarray(x)
        got = (sf.convolve if conv else sf.correlate)(w, mode="mirror", origin=origin).get()
        # reference: the periodic continuation along axis 0 made explicit (the dense filters take ONE mode), cropped
        xp = np.concatenate([x[-4:], x, x[:4]])
        ref = (orc.convolve if conv else orc.correlate)(xp, w, mode="mirror", origin=origin)[4:4 + nz]
        assert maxnorm_rel(got, ref) <= 1e-6, (conv, origin)
        if lo > 0:
            thin = SlabFilter(SlabPlan.self_loop(nz, lo - 1, hi), x.shape[1:], np.float32, self_comm)
            with pytest.raises(ValueError):
                (thin.convolve if conv else thin.correlate)(w, mode="mirror", origin=origin)
    u = rng.integers(0, 200, size=(nz, 24, 64)).astype(np.uint8)
    fp = rng.random((5, 3, 3)) > 0.3
    fp[0, 1, 1] = fp[4, 1, 1] = True
    st = rng.integers(0, 9, size=(3, 3, 3)).astype(np.float64)
    sf = SlabFilter(SlabPlan.self_loop(nz, 3, 3), u.shape[1:], np.uint8, self_comm)
    sf.local_in[...] = gpu.asarray(u)
    def periodic(fn):                        # the closed chain = the periodic continuation along axis 0, made explicit
        up = np.concatenate([u[-4:], u, u[:4]])
        return fn(up)[4:4 + nz]
    assert np.array_equal(sf.minimum_filter(footprint=fp, mode="nearest").get(),
                          periodic(lambda a: orc.minimum_filter(a, footprint=fp, mode="nearest")))
    assert np.array_equal(sf.maximum_filter(footprint=fp, mode="nearest", origin=(-1, 0, 0)).get(),
                          periodic(lambda a: orc.maximum_filter(a, footprint=fp, mode="nearest", origin=(-1, 0, 0))))
    assert np.array_equal(sf.grey_dilation(footprint=fp, mode="nearest", origin=(1, 0, 0)).get(),
                          periodic(lambda a: orc.grey_dilation(a, footprint=fp, mode="nearest", origin=(1, 0, 0))))
    assert np.array_equal(sf.grey_erosion(structure=st, mode="nearest").get(),
                          periodic(lambda a: orc.grey_erosion(a, structure=st, mode="nearest")))
    with pytest.raises(ValueError):
        sf.minimum_filter(footprint=np.ones((9, 1, 1), bool))       # reach 4 > halo 3


def test_sharded_interpolation_on_virtual_ranks(gpu, ndi):
    """ShardedInterp with a replicated input: the output planes of four virtual ranks, computed one after the other on
    this GPU (offset + M[:, 0] z0, the pre-image planes as a view), tile the unsplit call to float32 rounding (the
    coordinates are re-based: measured <= 1 ulp); map_coordinates is bit-identical; the pre-image of every rank is far
    smaller than the volume for a near-identity warp.  More than one real rank: tests/helpers/multirank_check.py."""
    from cupyimg_amd.distributed import ShardedInterp, SlabPlan, affine_axis0_range, preimage_planes
    rng = np.random.default_rng(80)
    x = rng.standard_normal((96, 64, 128)).astype(np.float32)
    xd = gpu.asarray(x)
    ang = np.deg2rad(7.0)
    M = np.diag([1.02, 1.0, 1.0]) @ np.array([[1, 0, 0], [0, np.cos(ang), -np.sin(ang)], [0, np.sin(ang), np.cos(ang)]])
    off = np.array([0.5, -1.25, 2.0])
    oshape = (90, 64, 128)
    for order, mode in [(1, "constant"), (0, "nearest"), (1, "mirror"), (3, "constant")]:
        full = ndi.affine_transform(xd, M, off, output_shape=oshape, order=order, mode=mode, cval=0.5).get()
        parts = []
        for r in range(4):
            plan = SlabPlan(oshape[0], 4, r, 0, 0)
            parts.append(ShardedInterp(plan).affine_transform(xd, M, off, output_shape=oshape, order=order, mode=mode, cval=0.5).get())
            if order <= 1 and mode == "constant":
                a, b = preimage_planes(*affine_axis0_range(M, off, oshape, plan.z0, plan.z1), x.shape[0], order, mode)
                assert b - a <= plan.n_local * 1.02 + 4
        got = np.concatenate(parts)
        assert np.allclose(got, full, rtol=0, atol=(2e-5 if order == 3 else 1e-6) * max(1.0, np.abs(full).max())), (order, mode)
    idx = np.indices(oshape).reshape(3, -1).astype(np.float64)
    coords = (M @ idx + off[:, None]).reshape((3,) + oshape).astype(np.float32)
    cd = gpu.asarray(coords)
    full = ndi.map_coordinates(xd, cd, order=1, mode="constant").get()
    parts = [ShardedInterp(SlabPlan(oshape[0], 4, r, 0, 0)).map_coordinates(
        xd, gpu.asarray(np.ascontiguousarray(coords[:, SlabPlan(oshape[0], 4, r, 0, 0).z0:SlabPlan(oshape[0], 4, r, 0, 0).z1])),
        order=1, mode="constant").get() for r in range(4)]
    assert np.array_equal(np.concatenate(parts), full)
    # the gather path on ONE rank ("distributed" input, no peers): coordinates that read planes 30 .. 60 only -- the
    # pre-image planes are copied out, axis 0 of the coordinates is re-based by an integer: bit-identical
    sub = np.ascontiguousarray(coords[:, 31:58])
    sub[0] = np.clip(sub[0], 30.25, 59.5)
    one = ShardedInterp(SlabPlan(sub.shape[1], 1, 0, 0, 0), in_plan=SlabPlan(x.shape[0], 1, 0, 0, 0))
    got = one.map_coordinates(xd, gpu.asarray(sub), order=1, mode="constant").get()
    assert np.array_equal(got, ndi.map_coordinates(xd, gpu.asarray(sub), order=1, mode="constant").get())
    Mz, offz = np.diag([0.25, 1.0, 1.0]), np.array([40.0, 0.0, 0.0])                 # output planes 0 .. 31 read input planes 40 .. 48
    one = ShardedInterp(SlabPlan(32, 1, 0, 0, 0), in_plan=SlabPlan(x.shape[0], 1, 0, 0, 0))
    got = one.affine_transform(xd, Mz, offz, output_shape=(32, 64, 128), order=1, mode="constant").get()
    assert np.array_equal(got, ndi.affine_transform(xd, Mz, offz, output_shape=(32, 64, 128), order=1, mode="constant").get())
