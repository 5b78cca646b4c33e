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
