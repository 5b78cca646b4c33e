"""Mvoxels/s and %HBM-roofline for every BASELINE.json config that fits one GPU
(config 0 is the CPU plumbing case, config 4 the 8-GPU case).  Prints one JSON
line per config.  Algorithmic bytes per voxel: SURVEY.md section 8(d).

    python scripts/bench_configs.py [--reps N] [--only H,B,C,D,Daff]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi

PEAK = 8000.0


def timeit(fn, reps):
    for _ in range(10):
        fn()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    ca.synchronize()
    return e0.elapsed_ms(e1) / reps / 1e3


def report(name, workload, voxels, alg_bytes_per_voxel, secs, extra=None):
    gbs = voxels * alg_bytes_per_voxel / secs / 1e9
    line = {"config": name, "workload": workload, "Mvoxels_per_s": round(voxels / secs / 1e6, 1),
            "ms": round(secs * 1e3, 4), "alg_bytes_per_voxel": alg_bytes_per_voxel,
            "achieved_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 4)}
    if extra:
        line.update(extra)
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="H,B,C,D,Daff,E")
    a = ap.parse_args()
    only = set(a.only.split(","))
    rng = np.random.default_rng(0)
    n = 512
    if only & {"H", "B", "D", "Daff"}:
        x = rng.standard_normal((n, n, n), dtype=np.float32)
        xd = ca.asarray(x)
        out = ca.empty(xd.shape, np.float32)
    if "H" in only:
        t = timeit(lambda: ndi.uniform_filter(xd, size=5, output=out), a.reps)
        report("H", "uniform_filter size=5, 512^3 float32", n ** 3, 8, t)
    if "B" in only:
        t = timeit(lambda: ndi.gaussian_filter(xd, sigma=2, output=out), a.reps)
        report("B", "gaussian_filter sigma=2 (17 taps/axis), 512^3 float32", n ** 3, 8, t)
    if "D" in only or "Daff" in only:
        ang = np.deg2rad(7.0)
        R = np.array([[1, 0, 0], [0, np.cos(ang), -np.sin(ang)], [0, np.sin(ang), np.cos(ang)]])
        M = np.diag([1.02, 1.0, 1.0]) @ R
        ctr = (n - 1) / 2.0
        off = ctr - M @ np.array([ctr] * 3) + np.array([0.5, -1.25, 2.0])
        if "Daff" in only:
            t = timeit(lambda: ndi.affine_transform(xd, M, off, order=1, mode="constant", output=out), a.reps)
            report("D-affine", "affine_transform order=1 3-D warp, 512^3 float32", n ** 3, 8, t)
        if "D" in only:
            idx = np.indices((n, n, n), dtype=np.float32).reshape(3, -1)
            coords = (M.astype(np.float32) @ idx + off.astype(np.float32)[:, None]).reshape(3, n, n, n)
            del idx
            cd = ca.asarray(coords)
            del coords
            t = timeit(lambda: ndi.map_coordinates(xd, cd, order=1, mode="constant", output=out), a.reps)
            report("D", "map_coordinates order=1 3-D affine warp, 512^3 float32 (+1.5 GiB coords)", n ** 3, 20, t)
            del cd
    if "E" in only:
        # one rank's share of config E: 2048^3 split over 8 GPUs = 256 planes + 4 halo planes each side
        xd = out = None
        ca.free_all_blocks()
        shape = (264, 2048, 2048)
        ed = ca.asarray(np.random.default_rng(2).standard_normal(shape, dtype=np.float32))
        eo = ca.empty(shape, np.float32)
        t = timeit(lambda: ndi.uniform_filter(ed, size=9, output=eo), max(3, a.reps // 2))
        report("E-slab", "uniform_filter size=9 on one rank's 264x2048x2048 float32 slab of the 2048^3 volume",
               shape[0] * shape[1] * shape[2], 8, t)
        ed = eo = None
        ca.free_all_blocks()
    if "C" in only:
        xd = out = None
        ca.free_all_blocks()
        m = 1024
        u = np.random.default_rng(1).integers(0, 256, size=(m, m, m), dtype=np.uint8)
        ud = ca.asarray(u)
        uo = ca.empty(ud.shape, np.uint8)
        t = timeit(lambda: ndi.grey_erosion(ud, size=7, output=uo), max(3, a.reps // 2))
        report("C", "grey_erosion size=7, 1024^3 uint8", m ** 3, 2, t)


if __name__ == "__main__":
    main()
