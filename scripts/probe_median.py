"""r5: rank filters on volumes / images -- the sizes people call them with; the 3 x 3 x 3 median on the kernel that shares its sorting
between windows (default) and on the per-voxel network (mi_debug_set_median27(0)).  One JSON line per call -> profiles/r5_rank_filters.txt
usage: python scripts/probe_median.py"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit
for n, dt in ((256, np.float32), (256, np.uint8), (512, np.float32), (512, np.uint8), (512, np.int16)):
    x = (np.random.default_rng(0).standard_normal((n,) * 3) * 50).astype(dt)
    xd = ca.asarray(x); out = ca.empty(x.shape, dt)
    for name, fn in (("median_filter 3", lambda: ndi.median_filter(xd, size=3, output=out)),
                     ("median_filter 5", lambda: ndi.median_filter(xd, size=5, output=out)),
                     ("percentile_filter 30 size 3", lambda: ndi.percentile_filter(xd, 30, size=3, output=out)),
                     ("rank_filter 3 size (1,3,3)", lambda: ndi.rank_filter(xd, 3, size=(1, 3, 3), output=out))):
        t, _ = timeit(fn, 3)
        if name == "median_filter 3":
            k27 = last_kernel()[4:60]
            _lib.load().mi_debug_set_median27(0)
            t0, _ = timeit(fn, 3)
            _lib.load().mi_debug_set_median27(1)
            print(json.dumps({"n": n, "dtype": np.dtype(dt).name, "call": name, "us": round(t * 1e6, 1), "of 8 TB/s": round(2 * x.nbytes / 8e12 / t, 4), "kernel": k27,
                              "per-voxel network us": round(t0 * 1e6, 1)}), flush=True)
            continue
        print(json.dumps({"n": n, "dtype": np.dtype(dt).name, "call": name, "us": round(t * 1e6, 1), "of 8 TB/s": round(2 * x.nbytes / 8e12 / t, 4),
                          "kernel": last_kernel()[4:60]}), flush=True)
img = (np.random.default_rng(1).standard_normal((4096, 4096)) * 50).astype(np.float32)
xd = ca.asarray(img); out = ca.empty(img.shape, np.float32)
for size in (3, 5, 7, 9, 11):
    t, _ = timeit(lambda: ndi.median_filter(xd, size=size, output=out), 3)
    print(json.dumps({"image": img.shape, "call": "median_filter %d" % size, "us": round(t * 1e6, 1), "of 8 TB/s": round(2 * img.nbytes / 8e12 / t, 4),
                      "kernel": last_kernel()[4:60]}), flush=True)
