"""N > 1 path on CPU: two (and four) processes over gloo run the halo exchange
schedule of SlabPlan (the same send/recv pairing, in the same order, that
mi_halo_exchange issues over RCCL on the GPU) with host buffers, filter their
extended slab with the CPU oracle and must reproduce the unsplit result bit for
bit -- uneven slabs, the closed chain of `wrap`, asymmetric origins, and the
two schedules of iterated binary morphology (SlabFilter.binary_erosion: one
exchange of iterations x reach planes; until-stable with a one-int OR)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, mode, size, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from cupyimg_amd.distributed import SlabPlan, halo_widths
    from oracle import ndimage as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)
    x = rng.standard_normal((20, 6, 8)).astype(np.float32)     # same volume on every rank
    lo, hi = halo_widths(size)
    plan = SlabPlan(x.shape[0], world, rank, lo, hi, wrap=(mode == "wrap"))
    ext = np.zeros((plan.n_ext,) + x.shape[1:], np.float32)
    ext[plan.local_slice] = x[plan.z0:plan.z1]

    # same pairing / order as csrc/halo.hip: "downwards" first, then "upwards"
    ops, keep = [], []
    def send(sl, peer):
        t = torch.from_numpy(np.ascontiguousarray(ext[sl])); keep.append(t)
        ops.append(dist.P2POp(dist.isend, t, peer))
    def recv(sl, peer):
        t = torch.empty(ext[sl].shape, dtype=torch.float32); keep.append((sl, t))
        ops.append(dist.P2POp(dist.irecv, t, peer))
    if plan.hi:
        if plan.prev >= 0: send(plan.send_to_prev(), plan.prev)
        if plan.next >= 0: recv(plan.recv_from_next(), plan.next)
    if plan.lo:
        if plan.next >= 0: send(plan.send_to_next(), plan.next)
        if plan.prev >= 0: recv(plan.recv_from_prev(), plan.prev)
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for item in keep:
        if isinstance(item, tuple):
            ext[item[0]] = item[1].numpy()

    assert np.array_equal(ext, x[plan.global_planes_of_ext()])      # halos carry the right planes
    res = orc.uniform_filter(ext, size, mode=mode)[plan.local_slice]
    ref = orc.uniform_filter(x, size, mode=mode)[plan.z0:plan.z1]
    ok = bool(np.array_equal(res, ref))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("mode,size", [("reflect", 5), ("wrap", 5), ("constant", 4)])
def test_two_rank_halo_exchange_gloo(mode, size):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, size, q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    results = sorted(q.get(timeout=5) for _ in range(2))
    assert results == [(0, True), (1, True)]
    for p in procs:
        assert p.exitcode == 0


def _exchange(dist, torch, ext, plan):
    """the pairing / order of csrc/halo.hip on host buffers"""
    ops, keep = [], []
    def send(sl, peer):
        t = torch.from_numpy(np.ascontiguousarray(ext[sl])); keep.append(t)
        ops.append(dist.P2POp(dist.isend, t, peer))
    def recv(sl, peer):
        t = torch.from_numpy(np.empty(ext[sl].shape, ext.dtype)); keep.append((sl, t))
        ops.append(dist.P2POp(dist.irecv, t, peer))
    if plan.hi:
        if plan.prev >= 0: send(plan.send_to_prev(), plan.prev)
        if plan.next >= 0: recv(plan.recv_from_next(), plan.next)
    if plan.lo:
        if plan.next >= 0: send(plan.send_to_next(), plan.next)
        if plan.prev >= 0: recv(plan.recv_from_prev(), plan.prev)
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for item in keep:
        if isinstance(item, tuple):
            ext[item[0]] = item[1].numpy()


def _worker4(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from cupyimg_amd.distributed import SlabPlan, halo_widths
    from oracle import ndimage as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(11)
    nz = 22                                                    # 6 + 6 + 5 + 5 planes: uneven slabs
    x = rng.standard_normal((nz, 7, 9)).astype(np.float32)
    fails = []
    # separable filter, every mode family, symmetric and asymmetric origins on axis 0
    for mode, size, origin in [("reflect", 5, 0), ("wrap", 5, 0), ("mirror", 4, -1), ("nearest", 5, 1), ("wrap", 3, -1),
                               ("constant", 7, 2)]:
        lo, hi = halo_widths(size, origin)
        plan = SlabPlan(nz, world, rank, lo, hi, wrap=(mode == "wrap"))
        ext = np.zeros((plan.n_ext,) + x.shape[1:], np.float32)
        ext[plan.local_slice] = x[plan.z0:plan.z1]
        _exchange(dist, torch, ext, plan)
        if not np.array_equal(ext, x[plan.global_planes_of_ext()]):
            fails.append(("halo", mode, size, origin))
        org = (origin, 0, 0)
        res = orc.uniform_filter(ext, size, mode=mode, origin=org)[plan.local_slice]
        ref = orc.uniform_filter(x, size, mode=mode, origin=org)[plan.z0:plan.z1]
        if not np.array_equal(res, ref):
            fails.append(("uniform", mode, size, origin))
    # iterated binary erosion: ONE exchange of iterations x reach planes
    b = rng.random((nz, 9, 10)) > 0.25
    st = orc.generate_binary_structure(3, 2)
    for iterations, border in [(1, 0), (3, 0), (2, 1)]:
        plan = SlabPlan(nz, world, rank, iterations, iterations)
        ext = np.zeros((plan.n_ext,) + b.shape[1:], np.uint8)
        ext[plan.local_slice] = b[plan.z0:plan.z1]
        _exchange(dist, torch, ext, plan)
        res = orc.binary_erosion(ext.astype(bool), structure=st, iterations=iterations, border_value=border)[plan.local_slice]
        ref = orc.binary_erosion(b, structure=st, iterations=iterations, border_value=border)[plan.z0:plan.z1]
        if not np.array_equal(res, ref):
            fails.append(("binary", iterations, border))
    # until stable: one iteration per exchange, the per-rank "changed" flags OR-ed (the one-int reduction)
    plan = SlabPlan(nz, world, rank, 1, 1)
    cur = np.zeros((plan.n_ext,) + b.shape[1:], np.uint8)
    cur[plan.local_slice] = b[plan.z0:plan.z1]
    steps = 0
    while True:
        _exchange(dist, torch, cur, plan)
        nxt = orc.binary_erosion(cur.astype(bool), structure=st, iterations=1, border_value=1)
        flag = torch.tensor([int(not np.array_equal(nxt[plan.local_slice], cur[plan.local_slice].astype(bool)))])
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        steps += 1
        if not int(flag[0]):
            break
        cur[plan.local_slice] = nxt[plan.local_slice]
    ref = orc.binary_erosion(b, structure=st, iterations=0, border_value=1)
    if not np.array_equal(cur[plan.local_slice].astype(bool), ref[plan.z0:plan.z1]):
        fails.append(("binary until stable", steps))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, fails))


def test_four_rank_uneven_slabs_origins_and_binary_iterations_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 4
    procs = [ctx.Process(target=_worker4, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(240) for p in procs]
    results = sorted(q.get(timeout=5) for _ in range(world))
    assert results == [(r, []) for r in range(world)], results
    for p in procs:
        assert p.exitcode == 0


# ---------------------------------------------------------------------------------------------------------------------
# r4: output-sharded interpolation (ShardedInterp) and the halo rule of dense / footprint filters, 4 gloo ranks on CPU
# ---------------------------------------------------------------------------------------------------------------------
class _HostComm:
    """distributed.HaloComm.sendrecv on host arrays over gloo: same op list, same per-pair order."""

    def __init__(self, dist, torch):
        self.dist, self.torch = dist, torch

    def sendrecv(self, ops):
        dist, torch = self.dist, self.torch
        p2p, recvs = [], []
        for a, peer, snd in ops:
            if snd:
                p2p.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(a)), peer))
            else:
                t = torch.from_numpy(np.empty(a.shape, a.dtype))
                recvs.append((a, t))
                p2p.append(dist.P2POp(dist.irecv, t, peer))
        if p2p:
            for w in dist.batch_isend_irecv(p2p):
                w.wait()
        for a, t in recvs:
            a[...] = t.numpy()


def _worker_interp(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from cupyimg_amd.distributed import ShardedInterp, SlabPlan, halo_widths
    from oracle import ndimage as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class HostInterp(ShardedInterp):                       # the device touch points replaced by NumPy + the CPU oracle
        def _alloc(self, shape, dtype):
            return np.empty(shape, dtype)

        def _run_affine(self, src, m, off, shape, output, order, mode, cval, prefilter):
            return orc.affine_transform(src, m, off, output_shape=shape, order=order, mode=mode, cval=cval)

        def _run_map(self, src, coordinates, output, order, mode, cval, prefilter):
            return orc.map_coordinates(src, coordinates, order=order, mode=mode, cval=cval)

        def _axis0_min_max(self, coordinates):
            return float(coordinates[0].min()), float(coordinates[0].max())

        def _shift_axis0(self, coordinates, a):
            c = coordinates.copy()
            c[0] -= a
            return c

    def allgather(vals):
        box = [None] * world
        dist.all_gather_object(box, list(vals))
        return box

    rng = np.random.default_rng(31)
    nz = 23                                                      # input planes 6 + 6 + 6 + 5
    x = rng.standard_normal((nz, 9, 11)).astype(np.float32)
    oshape = (26, 10, 12)                                        # output planes 7 + 7 + 6 + 6
    in_plan = SlabPlan(nz, world, rank, 0, 0)
    out_plan = SlabPlan(oshape[0], world, rank, 0, 0)
    comm = _HostComm(dist, torch)
    fails = []
    ang = np.deg2rad(9.0)
    mats = [
        (np.diag([1.02, 1.0, 1.0]) @ np.array([[1, 0, 0], [0, np.cos(ang), -np.sin(ang)], [0, np.sin(ang), np.cos(ang)]]), np.array([0.5, -1.25, 2.0])),
        (np.array([[0.8, 0.1, -0.05], [0.02, 1.0, 0.0], [0.0, 0.03, 0.95]]), np.array([-3.0, 0.4, 0.2])),     # axis 0 mixes with y, x; reaches below plane 0
        (np.diag([-1.0, 1.0, 1.0]), np.array([22.0, 0.0, 0.0])),                                              # flip: ranks need the far end
        (np.diag([0.3, 1.0, 1.0]), np.array([40.37, 0.0, 0.0])),                                              # everything above the volume
    ]
    for order in (0, 1):
        for mode in ("constant", "nearest", "reflect", "wrap"):
            for mi, (M, off) in enumerate(mats):
                ref = orc.affine_transform(x, M, off, output_shape=oshape, order=order, mode=mode, cval=0.25)[out_plan.z0:out_plan.z1]
                rep = HostInterp(out_plan).affine_transform(x, M, off, output_shape=oshape, order=order, mode=mode, cval=0.25)
                dis = HostInterp(out_plan, comm, in_plan).affine_transform(x[in_plan.z0:in_plan.z1], M, off, output_shape=oshape,
                                                                           order=order, mode=mode, cval=0.25)
                # a rank evaluates output plane z0 + k with the offset offset + M[:, 0] z0 (and axis 0 re-based to the
                # planes it holds): the same coordinates up to their last bit, so results agree to float32 rounding
                # (measured 3e-8), not bit for bit -- the two sharded forms agree exactly with each other
                tol = 1e-6 * max(1.0, float(np.abs(ref).max()))
                if not np.allclose(rep, ref, rtol=0, atol=tol):
                    fails.append(("affine replicated", order, mode, mi, float(np.abs(rep - ref).max())))
                if not np.array_equal(dis, rep):
                    fails.append(("affine distributed != replicated", order, mode, mi))
    # map_coordinates: every rank holds its slab of the coordinates; pre-image from their min / max, all-gathered
    idx = np.indices(oshape).reshape(3, -1).astype(np.float64)
    for mi, (M, off) in enumerate(mats[:3]):
        coords = (M @ idx + off[:, None]).reshape((3,) + oshape)
        coords += 0.3 * np.sin(coords)                           # not affine any more
        for mode in ("constant", "nearest", "mirror"):
            ref = orc.map_coordinates(x, coords, order=1, mode=mode, cval=-1.0)[out_plan.z0:out_plan.z1]
            mine = np.ascontiguousarray(coords[:, out_plan.z0:out_plan.z1])
            rep = HostInterp(out_plan).map_coordinates(x, mine, order=1, mode=mode, cval=-1.0)
            dis = HostInterp(out_plan, comm, in_plan, allgather).map_coordinates(x[in_plan.z0:in_plan.z1], mine, order=1, mode=mode, cval=-1.0)
            if not np.array_equal(rep, ref):
                fails.append(("map replicated", mode, mi))
            if not np.array_equal(dis, ref):
                fails.append(("map distributed", mode, mi))
    # dense correlate / convolve and footprint min / max: the halo follows from the window's axis-0 extent and origin
    # (convolution and dilation mirror the window); the extended slab filtered with the oracle equals the unsplit result
    w = rng.standard_normal((4, 3, 5))
    fp = rng.random((5, 3, 3)) > 0.3
    fp[0, 1, 1] = fp[4, 1, 1] = True
    u = rng.integers(0, 200, size=(nz, 8, 10)).astype(np.uint8)
    for what, origin in [("correlate", (0, 0, 0)), ("correlate", (1, 0, -1)), ("convolve", (0, 0, 0)), ("convolve", (-2, 1, 0)),
                         ("min_fp", (0, 0, 0)), ("min_fp", (-1, 0, 0)), ("dil_fp", (1, 0, 0))]:
        n0 = w.shape[0] if what in ("correlate", "convolve") else fp.shape[0]
        mirrored = what in ("convolve", "dil_fp")
        o0 = origin[0] if not mirrored else -origin[0] - (1 if n0 % 2 == 0 else 0)
        lo, hi = halo_widths(n0, o0)
        plan = SlabPlan(nz, world, rank, lo, hi)
        src = x if what in ("correlate", "convolve") else u
        ext = np.zeros((plan.n_ext,) + src.shape[1:], src.dtype)
        ext[plan.local_slice] = src[plan.z0:plan.z1]
        _exchange(dist, torch, ext, plan)
        fn = {"correlate": lambda a: orc.correlate(a, w, mode="mirror", origin=origin),
              "convolve": lambda a: orc.convolve(a, w, mode="mirror", origin=origin),
              "min_fp": lambda a: orc.minimum_filter(a, footprint=fp, mode="nearest", origin=origin),
              "dil_fp": lambda a: orc.grey_dilation(a, footprint=fp, mode="nearest", origin=origin)}[what]
        if not np.array_equal(fn(ext)[plan.local_slice], fn(src)[plan.z0:plan.z1]):
            fails.append((what, origin))
        # one plane less of halo on the side that needs it must show (the rule is tight): only checked where lo > 0
        try:
            SlabPlan(nz, world, rank, max(lo - 1, 0), hi).check_reach(n0, o0)
            if lo > 0 and rank > 0:
                fails.append((what, origin, "reach check too lax"))
        except ValueError:
            pass
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, fails))


def test_four_rank_sharded_interpolation_and_dense_halos_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 4
    procs = [ctx.Process(target=_worker_interp, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    results = sorted(q.get(timeout=5) for _ in range(world))
    assert results == [(r, []) for r in range(world)], results
    for p in procs:
        assert p.exitcode == 0
