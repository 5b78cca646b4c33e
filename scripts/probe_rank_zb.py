"""r5: planes per workgroup of the sorting-network rank kernel (mi_debug_set_rank_zb).   usage: python scripts/probe_rank_zb.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit
lib = _lib.load()
for dt, shape in ((np.float32, (512, 512, 512)), (np.uint8, (512, 512, 512)), (np.float32, (181, 217, 181))):
    x = (np.random.default_rng(0).standard_normal(shape) * 50).astype(dt)
    xd = ca.asarray(x); out = ca.empty(shape, dt)
    for name, fn in (("median 3", lambda: ndi.median_filter(xd, size=3, output=out)),
                     ("rank 8 of 27", lambda: ndi.rank_filter(xd, 8, size=3, output=out)),
                     ("median (1,3,3)", lambda: ndi.median_filter(xd, size=(1, 3, 3), output=out)),
                     ("median 5", lambda: ndi.median_filter(xd, size=5, output=out))):
        row = {"shape": shape, "dtype": np.dtype(dt).name, "call": name}
        for zb in (1, 2, 4, 8, 16, 32, 0):
            lib.mi_debug_set_rank_zb(zb)
            t, _ = timeit(fn, 3)
            row["zb %d" % zb] = round(t * 1e6, 1)
        lib.mi_debug_set_rank_zb(0)
        print(json.dumps(row), flush=True)
    del xd, out
    ca.free_all_blocks()
