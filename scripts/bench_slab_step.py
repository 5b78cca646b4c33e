"""Per-rank step time of the slab schedules on ONE GPU: a one-rank RCCL
communicator exchanging with itself stands in for the neighbours (real RCCL
launch + copy; the link is not xGMI).  Emulates the N-rank share of a volume.

    python scripts/bench_slab_step.py [--ranks 8] [--size 5] [--side 512] [--planes P]

Every figure is the average of a burst of back-to-back steps with one host
synchronisation per burst (what bench.py issues), after ~40 ms of the same
steps (settled clocks), best of 3 bursts.  Schedules: plain / overlapped
(mi_slab_separable3d_f32) and the r4 pipelined schedule (mi_slab_pipe_*:
2 or 3 resident inputs, queued directly or replayed from a hipGraph).
Output: one JSON line -> profiles/r4_slab_step.txt.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def burst_us(ca, fn, steps, settle_ms=40.0, bursts=3):
    t_end = time.perf_counter() + settle_ms / 1e3
    while time.perf_counter() < t_end:
        fn(20)
    ca.synchronize()
    best = None
    e0, e1 = ca.Event(), ca.Event()
    for _ in range(bursts):
        e0.record()
        fn(steps)
        e1.record()
        e1.synchronize()
        t = e0.elapsed_ms(e1) / steps * 1e3
        best = t if best is None else min(best, t)
    return round(best, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--size", type=int, default=5)
    ap.add_argument("--side", type=int, default=512)
    ap.add_argument("--planes", type=int, default=0, help="local planes per rank (default side / ranks)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--graphs", type=int, default=0, help="1: also time the hipGraph replays of every pipelined variant")
    ap.add_argument("--single-gpu-us", type=float, default=0.0, help="single-GPU step of the whole volume (0: measure it when it fits)")
    a = ap.parse_args()
    import cupyimg_amd as ca
    from cupyimg_amd.distributed import HaloComm, SlabFilter, SlabPlan, halo_widths
    from cupyimg_amd.scipy import ndimage as ndi

    lo, hi = halo_widths(a.size)
    nz = a.planes or a.side // a.ranks
    plan = SlabPlan.self_loop(nz, lo, hi)
    comm = HaloComm(1, 0, lambda u: u)
    sf = SlabFilter(plan, (a.side, a.side), np.float32, comm)
    sf.autotune = False
    rng = np.random.default_rng(0)
    for z in range(0, nz, 16):
        sf.local_in[z:z + 16] = ca.asarray(rng.standard_normal((min(16, nz - z), a.side, a.side)).astype(np.float32))
    fn = lambda x, y: ndi.uniform_filter(x, size=a.size, output=y)   # noqa: E731
    res = {"ranks_emulated": a.ranks, "local_planes": nz, "ext_shape": list(sf.ext_in.shape), "halo": [lo, hi],
           "halo_MiB_per_direction": round(lo * a.side * a.side * 4 / 2 ** 20, 2), "steps_per_burst": a.steps,
           "device": ca.device_name()}

    def loop(step):
        return lambda n: [step() for _ in range(n)]

    res["exchange_only_us"] = burst_us(ca, loop(lambda: comm.exchange(sf.ext_in, plan)), a.steps)
    res["filter_only_us"] = burst_us(ca, loop(lambda: fn(sf.ext_in, sf.ext_out)), a.steps)
    res["plain_us"] = burst_us(ca, loop(lambda: sf.uniform_filter(a.size, overlap=False)), a.steps)
    res["overlapped_us"] = burst_us(ca, loop(lambda: sf.uniform_filter(a.size, overlap=True)), a.steps)
    ref = sf.uniform_filter(a.size, overlap=False).copy()
    import ctypes
    from cupyimg_amd import _lib
    prio = _lib.load().mi_debug_set_pipe_normal_priority
    prio.argtypes = [ctypes.c_int]
    # (resident inputs, comm stream priority)
    variants = [(2, "high"), (3, "high"), (3, "normal")]
    for nbuf, pr in variants:
        prio(1 if pr == "normal" else 0)
        pipe = sf.uniform_pipeline(a.size, nbuf=nbuf)
        for k in range(1, nbuf):
            pipe.inputs[k][...] = sf.ext_in
        key = "pipelined_nbuf{}{}".format(nbuf, "_normalprio" if pr == "normal" else "")
        res[key + "_direct_us"] = burst_us(ca, lambda n: pipe.run(n, 0), a.steps)
        if a.graphs:
            try:
                res[key + "_graph_us"] = burst_us(ca, lambda n: pipe.run(n, 1), a.steps)
                per = 8 * nbuf
                res[key + "_graph8_us"] = burst_us(ca, lambda n: pipe.run(n, per), max(a.steps // per, 1) * per)
                res[key + "_info"] = pipe.info()
            except Exception as exc:             # a capture that fails inside RCCL: recorded, not fatal
                res[key + "_graph_error"] = repr(exc)[:200]
        ca.synchronize()
        res[key + "_bit_identical_to_plain"] = not ca.arrays_differ(pipe.local_out, ref)
        k_us, ex_us = pipe.measure()
        res[key + "_kernel_us"], res[key + "_exchange_us"] = round(k_us, 2), round(ex_us, 2)
        pipe.close()
        del pipe
    prio(0)
    single = a.single_gpu_us
    if not single and a.side ** 3 * 8 < 8 << 30:
        xd = ca.empty((a.side,) * 3, np.float32)
        xd.fill(1.0)
        od = ca.empty(xd.shape, np.float32)
        single = burst_us(ca, loop(lambda: ndi.uniform_filter(xd, size=a.size, output=od)), 50)
    res["single_gpu_whole_volume_us"] = single
    if single:
        best = min(v for k, v in res.items() if k.endswith("_us") and (k.startswith("pipelined") or k in ("plain_us", "overlapped_us"))
                   and not k.endswith(("kernel_us", "exchange_us")))
        res["best_step_us"] = best
        res["speedup_vs_single_gpu_at_{}_ranks".format(a.ranks)] = round(single / best, 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
