"""r5: float64 volumes (what nibabel's get_fdata() hands out): the common filters.   usage: python scripts/probe_f64.py"""
import sys, json
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit
for n in (256, 512):
    x = np.random.default_rng(0).standard_normal((n, n, n))
    xd = ca.asarray(x); out = ca.empty(x.shape, np.float64)
    for name, fn in (("median 3", lambda: ndi.median_filter(xd, size=3, output=out)), ("percentile 25 size 3", lambda: ndi.percentile_filter(xd, 25, size=3, output=out)), ("uniform 5", lambda: ndi.uniform_filter(xd, 5, output=out)),
                     ("gaussian 2", lambda: ndi.gaussian_filter(xd, 2.0, output=out)), ("max 5", lambda: ndi.maximum_filter(xd, 5, output=out))):
        t, _ = timeit(fn, 3)
        print(n, name, round(t * 1e6, 1), "us", round(2 * x.nbytes / 8e12 / t, 3), last_kernel()[4:60], flush=True)
    del xd, out; ca.free_all_blocks()
