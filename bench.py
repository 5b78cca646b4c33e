#!/usr/bin/env python3
"""Headline benchmark: uniform_filter(size=5) on a 512^3 float32 volume.

    python bench.py --gpus N --steps K --warmup W

N = 1: one fused HIP launch per step on a device-resident volume
(cupyimg_amd.scipy.ndimage.uniform_filter -> mi_separable3d_f32).
N > 1 (launched by torch.distributed.run, one rank per GPU): the same 512^3
volume is slab-partitioned on axis 0 (strong scaling); a step is one RCCL halo
exchange of the slab faces plus the fused filter on the rank's extended slab.
torch.distributed (gloo, CPU) is used for the rendezvous, the barriers and the
max-over-ranks reduction only -- all device work goes through libmi355img.

One JSON line is printed by rank 0.  `value` = voxels of the whole volume per
second (Mvoxels/s) with inputs already resident in HBM.  `roofline` prices the
fused kernel against 8.0 TB/s with the ALGORITHMIC 8 B/voxel (SURVEY.md
section 8d), timed live with HIP events on the library's stream.
`cpu_baseline` times the CPU oracle (scalar single-thread port) and
scipy.ndimage on the host cores of this box, on the full 512^3 volume, and the
oracle's output doubles as a full-size parity check of the GPU result.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SIDE = 512
SIZE = 5
HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy ceiling 6290
ALG_BYTES_PER_VOXEL = 8        # read 4 + write 4


def synth(shape, seed=0):
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def measured_traffic(world):
    """HBM bytes per launch from the rocprofv3 PMC passes of this same command
    (profiles/r2_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction
    applied).  Counters cannot be read from inside the process, so this is the
    committed measurement, valid for the single-GPU workload only."""
    if world != 1:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r2_traffic.json")) as f:
            return json.load(f)["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def _allcores_worker(args):
    """One slab of the slab-parallel SciPy run (child process of the CPU-only helper)."""
    name_in, name_out, shape, z0, z1, lo, hi = args
    from multiprocessing import shared_memory
    import scipy.ndimage as sndi
    a = shared_memory.SharedMemory(name=name_in)
    b = shared_memory.SharedMemory(name=name_out)
    x = np.ndarray(shape, np.float32, buffer=a.buf)
    y = np.ndarray(shape, np.float32, buffer=b.buf)
    e0, e1 = max(z0 - lo, 0), min(z1 + hi, shape[0])
    # at a global edge the slab edge is the volume edge, so `reflect` is evaluated exactly as unsplit
    y[z0:z1] = sndi.uniform_filter(x[e0:e1], size=SIZE)[z0 - e0:z0 - e0 + (z1 - z0)]
    a.close()
    b.close()
    return 0


def allcores_helper(nproc):
    """CPU-only helper (run as a child process, never touches the GPU):
    scipy.ndimage.uniform_filter on the same 512^3 volume, z-slabs over `nproc`
    processes sharing memory.  Prints one JSON line."""
    import multiprocessing as mp
    from multiprocessing import shared_memory
    x0 = synth((N_SIDE,) * 3)
    a = shared_memory.SharedMemory(create=True, size=x0.nbytes)
    b = shared_memory.SharedMemory(create=True, size=x0.nbytes)
    try:
        x = np.ndarray(x0.shape, np.float32, buffer=a.buf)
        x[...] = x0
        bounds = np.linspace(0, N_SIDE, nproc + 1).astype(int)
        jobs = [(a.name, b.name, x0.shape, int(bounds[i]), int(bounds[i + 1]), SIZE // 2, SIZE // 2)
                for i in range(nproc) if bounds[i + 1] > bounds[i]]
        best = None
        with mp.get_context("fork").Pool(nproc) as pool:
            pool.map(_allcores_worker, jobs)                       # warm-up (page faults, imports)
            for _ in range(2):
                t0 = time.perf_counter()
                pool.map(_allcores_worker, jobs)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
        print(json.dumps({"value": round(x0.size / best / 1e6, 1), "unit": "Mvoxels/s", "cores": nproc,
                          "note": "scipy.ndimage.uniform_filter on z-slabs (+2 halo planes) in %d processes, "
                                  "shared memory, best of 2" % nproc}))
    finally:
        a.close(); a.unlink(); b.close(); b.unlink()


def cpu_baseline(x, gpu_out):
    """Oracle (kind "port", 1 core) + scipy.ndimage on the same array."""
    from oracle import ndimage as orc
    orc.build()
    best = None
    ref = None
    for _ in range(2):
        t0 = time.perf_counter()
        ref = orc.uniform3d_f32(x, SIZE, "reflect")
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    res = {
        "value": round(x.size / best / 1e6, 2), "unit": "Mvoxels/s", "cores": 1, "kind": "port",
        "sample": "full 512^3 float32 volume, uniform_filter size=5 reflect, best of 2 "
                  "(oracle/ndimage_oracle.c orc_uniform3d_f32, scalar, single thread)",
        "host_cores_available": os.cpu_count(),
    }
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    res["cpu_model"] = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        import scipy.ndimage as sndi
        t0 = time.perf_counter()
        sref = sndi.uniform_filter(x, size=SIZE)
        dt = time.perf_counter() - t0
        res["scipy_ndimage"] = {"value": round(x.size / dt / 1e6, 2), "unit": "Mvoxels/s", "cores": 1,
                                "note": "scipy.ndimage.uniform_filter, single-threaded by construction, 1 run"}
        if gpu_out is not None:
            d = np.abs(gpu_out.astype(np.float64) - sref).max()
            res["parity_vs_scipy_maxnorm_rel"] = float(d / np.abs(sref).max())
    except Exception as exc:  # scipy missing on the box
        res["scipy_ndimage"] = {"error": repr(exc)}
    if gpu_out is not None:
        d = np.abs(gpu_out.astype(np.float64) - ref).max()
        res["parity_vs_oracle_maxnorm_rel"] = float(d / np.abs(ref).max())
    # all host cores: SciPy itself is single-threaded, so slabs in separate processes (a child process that
    # never touches the GPU)
    try:
        import subprocess
        nproc = max(1, min(os.cpu_count() or 1, 64))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-allcores-helper", str(nproc)],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=180)
        res["scipy_ndimage_all_cores"] = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as exc:
        res["scipy_ndimage_all_cores"] = {"error": repr(exc)[:200]}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-allcores-helper", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_allcores_helper:
        allcores_helper(args.cpu_allcores_helper)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus {} needs torch.distributed.run with {} ranks (WORLD_SIZE={})".format(
            args.gpus, args.gpus, world))

    import cupyimg_amd as ca
    from cupyimg_amd import distributed as dist_
    from cupyimg_amd.scipy import ndimage as ndi

    if not ca.is_available():
        raise SystemExit("bench.py needs an MI355X; no HIP device is visible")
    # one rank per GPU; on a box with fewer GPUs than ranks (functional dry runs only) ranks share devices
    ca.set_device(local_rank % max(ca.device_count(), 1))

    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        ca.synchronize()
        if dist is not None:
            dist.barrier()

    # ------------------------------------------------------------ workload
    x_host = synth((N_SIDE,) * 3) if rank == 0 or world > 1 else None
    if world == 1:
        xd = ca.asarray(x_host)
        out = ca.empty(xd.shape, np.float32)

        def step():
            ndi.uniform_filter(xd, size=SIZE, output=out)
    else:
        lo, hi = dist_.halo_widths(SIZE)
        plan = dist_.SlabPlan(N_SIDE, world, rank, lo, hi, wrap=False)

        def exchange_id(uid):
            box = [uid]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        comm = dist_.HaloComm(world, rank, exchange_id)
        sf = dist_.SlabFilter(plan, (N_SIDE, N_SIDE), np.float32, comm)
        sf.local_in[...] = ca.asarray(x_host[plan.z0:plan.z1])

        def step():
            sf.uniform_filter(SIZE)

    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = ca.Event(), ca.Event()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_ms(ev1)

    slab_ok = None
    if dist is not None:
        import torch
        # outside the timed region: every rank filters the whole volume on its own GPU and checks that its
        # slab of the distributed result is bit-identical to it
        full = ndi.uniform_filter(ca.asarray(x_host), size=SIZE)
        differ = ca.arrays_differ(sf.local_out, full[plan.z0:plan.z1])
        t = torch.tensor([elapsed, dev_ms, float(differ)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dev_ms, slab_ok = float(t[0]), float(t[1]), t[2].item() == 0.0

    voxels = N_SIDE ** 3
    ms_per_step = elapsed / args.steps * 1e3
    value = voxels / (elapsed / args.steps) / 1e6

    if rank == 0:
        kernel_s = dev_ms / 1e3 / args.steps          # HIP-event time per step on the launch stream
        per_gpu_voxels = voxels / world
        achieved = ALG_BYTES_PER_VOXEL * per_gpu_voxels / kernel_s / 1e9
        roofline = {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(world),
            "kernel": "mi::sep3d_lean_kernel<5,12,4,3,1,false> (fused x/z/y separable pass)",
            "alg_bytes_per_launch": ALG_BYTES_PER_VOXEL * per_gpu_voxels,
            "avg_launch_us": round(kernel_s * 1e6, 2),
        }
        if world == 1:
            # the practical ceiling of this box: a device-to-device copy of the same 512 MiB (read + write)
            for _ in range(3):
                out[...] = xd
            c0, c1 = ca.Event(), ca.Event()
            c0.record()
            for _ in range(10):
                out[...] = xd
            c1.record()
            ca.synchronize()
            copy_gbs = ALG_BYTES_PER_VOXEL * voxels / (c0.elapsed_ms(c1) / 10 / 1e3) / 1e9
            roofline["d2d_copy_GBps_same_bytes"] = round(copy_gbs, 1)
            roofline["frac_of_d2d_copy"] = round(achieved / copy_gbs, 4)
            step()                                        # `out` holds the filter result again (parity leg below)
            ca.synchronize()
        if world == 1 and not args.no_cpu:
            cpu = cpu_baseline(x_host, out.get())
        else:
            cpu = None
        line = {
            "metric": "Mvoxels/s, uniform_filter size=5 on 512^3 float32",
            "value": round(value, 1), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "uniform_filter size=5 mode=reflect on 512x512x512 float32, device resident",
                       "partition": "z-slabs x{} + RCCL halo exchange overlapped with the interior planes".format(world) if world > 1 else "single GPU",
                       "device": ca.device_name()},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if slab_ok is not None:
            line["slabs_bit_identical_to_single_gpu"] = slab_ok
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
