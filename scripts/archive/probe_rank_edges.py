"""r5: how much of the sorting-network rank kernels' time is the boundary path?  The same voxel count with 12 % of the waves
touching an edge (rows of 2048) and with every wave touching one (rows of 128).   usage: python scripts/probe_rank_edges.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit
for dt in (np.float32, np.uint8):
    for shape in ((128, 128, 2048), (2048, 128, 128), (512, 512, 512)):
        x = (np.random.default_rng(0).standard_normal(shape) * 50).astype(dt)
        xd = ca.asarray(x); out = ca.empty(shape, dt)
        for name, fn in (("median 3", lambda: ndi.median_filter(xd, size=3, output=out)),
                         ("rank 8 of 27", lambda: ndi.rank_filter(xd, 8, size=3, output=out)),
                         ("median (1,5,5)", lambda: ndi.median_filter(xd, size=(1, 5, 5), output=out)),
                         ("median 5", lambda: ndi.median_filter(xd, size=5, output=out))):
            t, _ = timeit(fn, 3)
            print(json.dumps({"shape": shape, "dtype": np.dtype(dt).name, "call": name, "us": round(t * 1e6, 1), "ns/voxel": round(t * 1e9 / x.size, 4)}), flush=True)
        del xd, out
        ca.free_all_blocks()
