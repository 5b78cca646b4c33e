"""Per-rank step time of the slab schedule on ONE GPU: a one-rank RCCL
communicator exchanging with itself stands in for the neighbours, so the RCCL
launch + copy latency is real while the link is not xGMI.  Emulates the
N-rank share of the 512^3 headline volume.

    python scripts/bench_slab_step.py [--ranks 8] [--size 5] [--side 512]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--size", type=int, default=5)
    ap.add_argument("--side", type=int, default=512)
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    import cupyimg_amd as ca
    from cupyimg_amd.distributed import HaloComm, SlabFilter, halo_widths
    from cupyimg_amd.scipy import ndimage as ndi
    from test_gpu_halo import _SelfLoopPlan

    lo, hi = halo_widths(a.size)
    nz = a.side // a.ranks
    plan = _SelfLoopPlan(nz, lo, hi)
    comm = HaloComm(1, 0, lambda u: u)
    sf = SlabFilter(plan, (a.side, a.side), np.float32, comm)
    sf.local_in[...] = ca.asarray(np.random.default_rng(0).standard_normal((nz, a.side, a.side)).astype(np.float32))
    fn = lambda x, y: ndi.uniform_filter(x, size=a.size, output=y)   # noqa: E731
    res = {"ranks_emulated": a.ranks, "local_planes": nz, "halo": [lo, hi]}
    for name, step in [("exchange_only", lambda: comm.exchange(sf.ext_in, plan)),
                       ("filter_only", lambda: fn(sf.ext_in, sf.ext_out)),
                       ("plain", lambda: sf.step(fn)), ("overlapped", lambda: sf.step_overlapped(fn)),
                       ("native_overlap", lambda: sf.uniform_filter(a.size, overlap=True)),
                       ("native", lambda: sf.uniform_filter(a.size))]:
        for _ in range(20):
            step()
        ca.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        ca.synchronize()
        res[name + "_us"] = round((time.perf_counter() - t0) / a.steps * 1e6, 1)
    res["speedup_vs_1gpu_205us"] = round(205.0 / res["native_us"], 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
