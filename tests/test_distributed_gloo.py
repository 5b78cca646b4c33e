"""N > 1 path on CPU: two processes over gloo run the halo exchange schedule of
SlabPlan (the same send/recv pairing, in the same order, that mi_halo_exchange
issues over RCCL on the GPU) with host buffers, filter their extended slab with
the CPU oracle and must reproduce the unsplit result bit for bit."""
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
