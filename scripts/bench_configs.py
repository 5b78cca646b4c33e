"""Mvoxels/s and %HBM-roofline for every BASELINE.json config that fits one GPU
(config 0 is the CPU plumbing case: scripts/bench_config_a.py; config 4's multi-rank
leg: bench.py --config E), each with a full-size parity field.  One JSON line per
config.  Algorithmic bytes per voxel: SURVEY.md section 8(d).

    python scripts/bench_configs.py [--reps N] [--only H,B,C,D,Daff,E] [--no-parity]

Timing: 10 warm launches, then `reps` (default 40) back-to-back launches between two
HIP events -- long enough that the clocks have settled to the power budget (DESIGN.md,
throttling note), i.e. the SUSTAINED number; `first_ms` is the mean of the first five
launches of a cold burst for comparison.  `parity` compares the result of the timed
call with scipy.ndimage on EVERY plane (r4: z sub-slabs that tile the volume, over the host
cores; tests/helpers/fullsize.py, the same checks as tests/test_gpu_baseline_full.py), outside
the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs

PEAK = 8000.0


def timeit(fn, reps):
    """sustained: seconds per call over `reps` back-to-back launches (at least 60 ms of them) after at least 30 ms of
    the same launches -- the clocks of these boxes need 10-20 ms of load to settle (a 0.4 ms kernel measured 10 % slower
    over launches 11-50 of a process than over launches 31-180: DESIGN.md section 0); first: the mean of the first five
    launches after half a second of idling."""
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    ca.synchronize()
    per_call = max(e0.elapsed_ms(e1) / 10, 1e-3)
    for _ in range(max(0, int(30.0 / per_call) - 10)):
        fn()
    reps = max(reps, int(60.0 / per_call))
    ca.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    ca.synchronize()
    sustained = e0.elapsed_ms(e1) / reps / 1e3
    time.sleep(0.5)                       # let the chip cool: the first launches of a burst run at full clocks
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    ca.synchronize()
    return sustained, e0.elapsed_ms(e1) / 5 / 1e3


def report(name, workload, voxels, alg_bytes_per_voxel, secs, first, extra=None):
    gbs = voxels * alg_bytes_per_voxel / secs / 1e9
    line = {"config": name, "workload": workload, "Mvoxels_per_s": round(voxels / secs / 1e6, 1),
            "ms": round(secs * 1e3, 4), "first_ms": round(first * 1e3, 4), "alg_bytes_per_voxel": alg_bytes_per_voxel,
            "achieved_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 4)}
    if extra:
        line.update(extra)
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--only", default="H,B,C,D,Daff,E")
    ap.add_argument("--no-parity", action="store_true")
    a = ap.parse_args()
    only = set(a.only.split(","))
    import scipy.ndimage as sndi
    par = not a.no_parity
    n = fs.N_H
    if only & {"H", "B", "D", "Daff"}:
        x = fs.volume_f32((n, n, n), seed=0)
        xd = ca.asarray(x)
        out = ca.empty(xd.shape, np.float32)
    if "H" in only:
        t, t1 = timeit(lambda: ndi.uniform_filter(xd, size=5, output=out), a.reps)
        p = {"parity": {"maxnorm_rel_vs_scipy": fs.whole_volume_filter(
            x, out.get(), 2, 2, lambda s: sndi.uniform_filter(s.astype(np.float64), size=5)), "planes": "all",
            "tol": 1e-6}} if par else None
        report("H", "uniform_filter size=5, 512^3 float32", n ** 3, 8, t, t1, p)
    if "B" in only:
        t, t1 = timeit(lambda: ndi.gaussian_filter(xd, sigma=2, output=out), a.reps)
        p = {"parity": {"maxnorm_rel_vs_scipy": fs.whole_volume_filter(
            x, out.get(), 8, 8, lambda s: sndi.gaussian_filter(s.astype(np.float64), sigma=2), planes=16), "planes": "all",
            "tol": 1e-6}} if par else None
        report("B", "gaussian_filter sigma=2 (17 taps/axis), 512^3 float32", n ** 3, 8, t, t1, p)
    if "D" in only or "Daff" in only:
        M, off = fs.affine_case(n)
        if "Daff" in only:
            t, t1 = timeit(lambda: ndi.affine_transform(xd, M, off, order=1, mode="constant", output=out), a.reps)
            p = {"parity": {"abs_err_over_max1_vs_scipy": fs.whole_volume_affine(x, M, off, out.get()), "planes": "all",
                            "tol": 2e-6}, "kernel": ca.last_kernel()[:90]} if par else {"kernel": ca.last_kernel()[:90]}
            report("D-affine", "affine_transform order=1 3-D warp, 512^3 float32", n ** 3, 8, t, t1, p)
        if "D" in only:
            coords = fs.affine_coords_f32(n)
            cd = ca.asarray(coords)
            t, t1 = timeit(lambda: ndi.map_coordinates(xd, cd, order=1, mode="constant", output=out), a.reps)
            p = {"parity": {"abs_err_over_max1_vs_scipy": fs.whole_volume_map_coordinates(x, coords, out.get()), "planes": "all",
                            "tol": 2e-6}} if par else None
            report("D", "map_coordinates order=1 3-D affine warp, 512^3 float32 (+1.5 GiB coords)", n ** 3, 20, t, t1, p)
            del cd, coords
    if "E" in only:
        # one rank's share of config E: 2048^3 split over 8 GPUs = 256 planes + 4 halo planes each side
        xd = out = None
        ca.free_all_blocks()
        shape = fs.E_SLAB
        xe = fs.slab_volume_f32(shape)
        ed = ca.asarray(xe)
        eo = ca.empty(shape, np.float32)
        t, t1 = timeit(lambda: ndi.uniform_filter(ed, size=9, output=eo), max(5, a.reps // 2))
        p = {"parity": {"maxnorm_rel_vs_scipy": fs.whole_volume_filter(
            xe, eo.get(), 4, 4, lambda s: sndi.uniform_filter(s.astype(np.float64), size=9), planes=4), "planes": "all",
            "tol": 1e-6}} if par else None
        report("E-slab", "uniform_filter size=9 on one rank's 264x2048x2048 float32 slab of the 2048^3 volume",
               shape[0] * shape[1] * shape[2], 8, t, t1, p)
        ed = eo = xe = None
        ca.free_all_blocks()
    if "C" in only:
        xd = out = None
        ca.free_all_blocks()
        m = fs.N_C
        u = fs.volume_u8((m, m, m), seed=1)
        ud = ca.asarray(u)
        uo = ca.empty(ud.shape, np.uint8)
        t, t1 = timeit(lambda: ndi.grey_erosion(ud, size=7, output=uo), max(5, a.reps // 2))
        p = {"parity": {"voxels_differing_from_scipy": fs.whole_volume_filter(
            u, uo.get(), 3, 3, lambda s: sndi.grey_erosion(s, size=7), exact=True, planes=16), "planes": "all",
            "tol": 0}} if par else None
        report("C", "grey_erosion size=7, 1024^3 uint8", m ** 3, 2, t, t1, p)


if __name__ == "__main__":
    main()
