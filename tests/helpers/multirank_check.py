"""One rank of a real multi-GPU slab run (tests/test_gpu_multirank.py launches N of these with
torch.distributed.run, one per GPU, BEFORE anything in them has touched a device).

Every rank holds one z-slab of a synthetic volume, exchanges halos with its neighbours over RCCL
(send/recv, xGMI) and checks on its own GPU that

  * every schedule of a step (plain, overlapped, pipelined with 2 / 3 resident inputs, hipGraph replay) gives the
    planes of the single-GPU filter of the WHOLE volume bit for bit (the rank filters the whole volume itself),
  * the planes around its seams agree with scipy.ndimage on the host (1e-6),
  * uneven slabs, `wrap` (closed chain), asymmetric origins, kernels without plane ranges (25 taps), uint8 grey erosion
    (plain and overlapped), iterated and until-stable binary erosion follow the same rule.

With WORLD_SIZE = 1 (no launcher) the same code runs without neighbours -- the single-GPU boxes of the pool use that
to keep this file honest.  Prints "MULTIRANK OK <world>" on rank 0; any failure exits non-zero on that rank
(torch.distributed.run then stops the others).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import cupyimg_amd as ca
    from cupyimg_amd import distributed as D
    from cupyimg_amd.scipy import ndimage as ndi
    import scipy.ndimage as sndi

    ca.set_device(local_rank)
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def exchange_id(uid):
        box = [uid]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def reduce_max(vals):
        if dist is None:
            return list(vals)
        t = torch.tensor(list(vals), dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    def any_changed(flag):
        return bool(reduce_max([1.0 if flag else 0.0])[0])

    comm = D.HaloComm(world, rank, exchange_id) if world > 1 else None
    checks = []

    def slab_filter(x, lo, hi, wrap=False):
        plan = D.SlabPlan(x.shape[0], world, rank, lo, hi, wrap=wrap)
        sf = D.SlabFilter(plan, x.shape[1:], x.dtype, comm, reduce_max=reduce_max)
        sf.local_in[...] = ca.asarray(x[plan.z0:plan.z1])
        return plan, sf

    def same(got, want, what):
        ok = np.array_equal(got, want)
        checks.append((what, ok))
        assert ok, "rank {}: {} differs from the single-GPU result ({} voxels)".format(rank, what, int((got != want).sum()))

    def seams(got_local, plan, ref_fn, x, reach, what, tol=1e-6):
        """planes next to this rank's slab faces against SciPy on a host block with `reach` planes of context"""
        nz = x.shape[0]
        worst = 0.0
        for a, b in ((plan.z0, min(plan.z0 + reach + 1, plan.z1)), (max(plan.z1 - reach - 1, plan.z0), plan.z1)):
            e0, e1 = max(a - reach, 0), min(b + reach, nz)
            ref = ref_fn(x[e0:e1].astype(np.float64))[a - e0:a - e0 + (b - a)]
            got = got_local[a - plan.z0:b - plan.z0].astype(np.float64)
            worst = max(worst, float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)))
        checks.append((what + " seams vs scipy {:.1e}".format(worst), worst <= tol))
        assert worst <= tol, "rank {}: {} seam parity {:.3e}".format(rank, what, worst)

    rng = np.random.default_rng(2026)

    # ---- 1. the headline filter on even slabs: every schedule, bit-identical to the whole-volume launch
    nz = 64 * world
    x = rng.standard_normal((nz, 160, 512)).astype(np.float32)
    full = ndi.uniform_filter(ca.asarray(x), size=5).get()
    plan, sf = slab_filter(x, 2, 2)
    want = full[plan.z0:plan.z1]
    same(sf.uniform_filter(5, overlap=False).get(), want, "uniform 5 plain")
    if world > 1:
        same(sf.uniform_filter(5, overlap=True).get(), want, "uniform 5 overlapped")
        sf.autotune = True
        sf.warm(lambda: sf.uniform_filter(5))
        st = sf.schedule_of("uniform")
        assert st is not None and st["agreed_across_ranks"], st
        choice = reduce_max([float(st["choice"]), -float(st["choice"])])
        assert choice[0] == -choice[1], "ranks disagree on the schedule"
        same(sf.uniform_filter(5).get(), want, "uniform 5 tuned schedule")
    for nbuf in (2, 3):
        sf.local_in[...] = ca.asarray(x[plan.z0:plan.z1])
        pipe = sf.uniform_pipeline(5, nbuf=nbuf)
        for k in range(1, nbuf):
            pipe.inputs[k][...] = sf.ext_in
        for graph in (0, 1, 2 * nbuf):
            sf.ext_out.fill(0.0)
            pipe.run(2 * nbuf + 1, graph)
            same(pipe.local_out.get(), want, "uniform 5 pipelined nbuf {} graph {}".format(nbuf, graph))
        # a sequence of DIFFERENT volumes through the pipeline: volume j is x + j
        outs = []
        depth = nbuf - 1
        for j in range(5 + depth):
            if j < 5:
                pipe.local_in(j % nbuf)[...] = ca.asarray(x[plan.z0:plan.z1] + np.float32(j))
                pipe.submit(j % nbuf)
            if j >= depth:
                outs.append(pipe.compute((j - depth) % nbuf).get())
        for j, o in enumerate(outs):
            wj = ndi.uniform_filter(ca.asarray(x + np.float32(j)), size=5).get()[plan.z0:plan.z1]
            same(o, wj, "pipelined sequence nbuf {} volume {}".format(nbuf, j))
        pipe.close()
    sf.local_in[...] = ca.asarray(x[plan.z0:plan.z1])
    seams(sf.uniform_filter(5, overlap=False).get(), plan, lambda v: sndi.uniform_filter(v, size=5), x, 2, "uniform 5")

    # ---- 2. uneven slabs, 9 taps with an origin along z, wrap (closed chain), constant
    nz = 50 * world + 3
    x = rng.standard_normal((nz, 40, 256)).astype(np.float32)
    for mode, origin in (("reflect", 0), ("wrap", 0), ("nearest", 1), ("mirror", -2)):
        w = np.full(9, 1.0 / 9)
        lo, hi = D.halo_widths(9, origin)
        plan, sf = slab_filter(x, lo, hi, wrap=(mode == "wrap"))
        for overlap in ((False, True) if world > 1 else (False,)):
            got = sf.separable([w, w, w], mode, 0.0, (origin, 0, 0), overlap=overlap).get()
            ref = sndi.correlate1d(sndi.correlate1d(sndi.correlate1d(x.astype(np.float64), w, 0, mode=mode, origin=origin), w, 1, mode=mode),
                                   w, 2, mode=mode)[plan.z0:plan.z1]
            err = float(np.abs(got - ref).max() / np.abs(ref).max())
            checks.append(("9 taps {} origin {} overlap {} vs scipy {:.1e}".format(mode, origin, overlap, err), err <= 1e-6))
            assert err <= 1e-6, (rank, mode, origin, overlap, err)

    # ---- 3. a kernel without plane ranges (25 taps): thin slabs, every rank refuses the overlapped form up front
    nz = 24 * world
    x = rng.standard_normal((nz, 24, 256)).astype(np.float32)
    lo, hi = D.halo_widths(25)
    plan, sf = slab_filter(x, lo, hi)
    ref = sndi.gaussian_filter(x.astype(np.float64), 3.0)[plan.z0:plan.z1]
    for overlap in (True, None, False):
        sf.autotune = overlap is None
        got = sf.gaussian_filter(3.0, overlap=overlap).get()
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
        checks.append(("gaussian 25 taps overlap {} vs scipy {:.1e}".format(overlap, err), err <= 1e-6))
        assert err <= 1e-6, (rank, overlap, err)
    pipe = sf.gaussian_pipeline(3.0, nbuf=2)
    pipe.inputs[1][...] = sf.ext_in
    got = pipe.run(3, 0).get()
    err = float(np.abs(got - ref).max() / np.abs(ref).max())
    assert not pipe.info()["planes_ok"] and err <= 1e-6, (rank, err)
    pipe.close()

    # ---- 4. uint8 grey erosion (config C's operation), plain and overlapped; binary erosion iterated / until stable
    nz = 40 * world + 1
    u = rng.integers(0, 256, size=(nz, 64, 128)).astype(np.uint8)
    plan, sf = slab_filter(u, 3, 3)
    want = sndi.grey_erosion(u, size=7)[plan.z0:plan.z1]
    same(sf.grey_erosion(7).get(), want, "grey_erosion 7 plain")
    if world > 1:
        same(sf.grey_erosion(7, overlap=True).get(), want, "grey_erosion 7 overlapped")
    b = rng.random((nz, 48, 64)) > 0.25
    plan, sf = slab_filter(b, 2, 2)
    same(sf.binary_erosion(iterations=2).get(), sndi.binary_erosion(b, iterations=2)[plan.z0:plan.z1], "binary_erosion x2")
    sf.local_in[...] = ca.asarray(b[plan.z0:plan.z1])
    same(sf.binary_erosion(iterations=0, any_changed=any_changed).get(), sndi.binary_erosion(b, iterations=0)[plan.z0:plan.z1],
         "binary_erosion until stable")

    # ---- 5. dense correlate with an even, shifted kernel; footprint minimum
    nz = 32 * world + 5
    x = rng.standard_normal((nz, 24, 64)).astype(np.float32)
    w = rng.standard_normal((4, 3, 5))
    for conv, origin in ((False, (1, 0, -1)), (True, (-2, 1, 0))):
        o0 = origin[0] if not conv else -origin[0] - 1
        lo, hi = D.halo_widths(4, o0)
        plan, sf = slab_filter(x, lo, hi)
        got = (sf.convolve if conv else sf.correlate)(w, mode="mirror", origin=origin).get()
        ref = (sndi.convolve if conv else sndi.correlate)(x.astype(np.float64), w, mode="mirror", origin=origin)[plan.z0:plan.z1]
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
        checks.append(("dense {} origin {} vs scipy {:.1e}".format("convolve" if conv else "correlate", origin, err), err <= 1e-6))
        assert err <= 1e-6, (rank, conv, origin, err)
    fp = rng.random((5, 3, 3)) > 0.3
    fp[0, 1, 1] = fp[4, 1, 1] = True
    u = rng.integers(0, 200, size=(nz, 24, 64)).astype(np.uint8)
    plan, sf = slab_filter(u, 2, 2)
    same(sf.minimum_filter(footprint=fp, mode="nearest").get(), sndi.minimum_filter(u, footprint=fp, mode="nearest")[plan.z0:plan.z1],
         "minimum_filter footprint")

    # ---- 6. output-sharded interpolation: replicated input and slab-distributed input (pre-image planes fetched from
    # their owners with one group of RCCL send / recv)
    def allgather(vals):
        if dist is None:
            return [list(vals)]
        box = [None] * world
        dist.all_gather_object(box, list(vals))
        return box

    nz = 40 * world + 3
    x = rng.standard_normal((nz, 48, 128)).astype(np.float32)
    oshape = (36 * world + 2, 48, 128)
    ang = np.deg2rad(7.0)
    mats = [(np.diag([1.02, 1.0, 1.0]) @ np.array([[1, 0, 0], [0, np.cos(ang), -np.sin(ang)], [0, np.sin(ang), np.cos(ang)]]),
             np.array([0.5, -1.25, 2.0])),
            (np.array([[0.9, 0.05, -0.02], [0.02, 1.0, 0.0], [0.0, 0.03, 0.95]]), np.array([-3.0, 0.4, 0.2])),
            (np.diag([-1.0, 1.0, 1.0]), np.array([nz - 1.0, 0.0, 0.0]))]                 # flip: every rank needs the far end
    in_plan = D.SlabPlan(nz, world, rank, 0, 0)
    out_plan = D.SlabPlan(oshape[0], world, rank, 0, 0)
    xd_full = ca.asarray(x)
    xd_mine = ca.asarray(x[in_plan.z0:in_plan.z1])
    for mi, (M, off) in enumerate(mats):
        for order, mode in ((1, "constant"), (0, "nearest"), (1, "reflect")):
            ref = sndi.affine_transform(x.astype(np.float64), M, off, output_shape=oshape, order=order, mode=mode, cval=0.25,
                                        prefilter=False)[out_plan.z0:out_plan.z1]
            rep = D.ShardedInterp(out_plan).affine_transform(xd_full, M, off, output_shape=oshape, order=order, mode=mode, cval=0.25).get()
            dis = D.ShardedInterp(out_plan, comm, in_plan).affine_transform(xd_mine, M, off, output_shape=oshape, order=order,
                                                                          mode=mode, cval=0.25).get()
            same(dis, rep, "sharded affine {} order {} {}: distributed input == replicated input".format(mi, order, mode))
            if order == 1:
                err = float(np.abs(rep - ref).max() / max(1.0, np.abs(ref).max()))
                checks.append(("sharded affine {} {} vs scipy {:.1e}".format(mi, mode, err), err <= 2e-6))
                assert err <= 2e-6, (rank, mi, mode, err)
    idx = np.indices((out_plan.n_local,) + oshape[1:]).reshape(3, -1).astype(np.float64)
    idx[0] += out_plan.z0
    M, off = mats[1]
    coords = (M @ idx + off[:, None]).reshape((3, out_plan.n_local) + oshape[1:])
    coords += 0.3 * np.sin(coords)
    cd = ca.asarray(coords.astype(np.float32))
    rep = D.ShardedInterp(out_plan).map_coordinates(xd_full, cd, order=1, mode="constant").get()
    dis = D.ShardedInterp(out_plan, comm, in_plan, allgather).map_coordinates(xd_mine, cd, order=1, mode="constant").get()
    same(dis, rep, "sharded map_coordinates: distributed input == replicated input")
    ref = sndi.map_coordinates(x.astype(np.float64), coords.astype(np.float32), order=1, mode="constant", prefilter=False)
    err = float(np.abs(rep - ref).max() / max(1.0, np.abs(ref).max()))
    assert err <= 2e-6, (rank, err)

    if dist is not None:
        dist.barrier()
    ca.synchronize()
    if comm is not None:
        comm.close()
    if rank == 0:
        for what, ok in checks:
            print(("ok   " if ok else "FAIL ") + what)
        print("MULTIRANK OK {}".format(world))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
