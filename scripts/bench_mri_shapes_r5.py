"""r5: the calls this round touched on the shapes neuroimaging volumes come in (MNI152 1 mm / 2 mm grids, 256^3 conformed,
a 4-D EPI frame): whole call, microseconds, fraction of 8 TB/s at one read + one write.  -> profiles/r5_mri_shapes.txt
usage: python scripts/bench_mri_shapes_r5.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit

rng = np.random.default_rng(0)
for shape in ((181, 217, 181), (91, 109, 91), (256, 256, 256), (64, 104, 104), (193, 229, 193)):
    for dt in (np.float32, np.float64):
        x = rng.standard_normal(shape).astype(dt)
        xd = ca.asarray(x); out = ca.empty(shape, dt)
        calls = [("uniform_filter 3", lambda: ndi.uniform_filter(xd, 3, output=out)),
                 ("uniform_filter 5", lambda: ndi.uniform_filter(xd, 5, output=out)),
                 ("gaussian_filter 1.0", lambda: ndi.gaussian_filter(xd, 1.0, output=out)),
                 ("gaussian_filter 2.0 constant", lambda: ndi.gaussian_filter(xd, 2.0, mode="constant", output=out)),
                 ("median_filter 3", lambda: ndi.median_filter(xd, size=3, output=out)),
                 ("percentile_filter 25 size 3", lambda: ndi.percentile_filter(xd, 25, size=3, output=out)),
                 ("grey_erosion 3", lambda: ndi.grey_erosion(xd, size=3, output=out))]
        if dt == np.float32:
            calls += [("rotate 7 (order 3)", lambda: ndi.rotate(xd, 7.0, reshape=False, output=out)),
                      ("zoom 1.0 -> shift (0.5, -0.25, 0.75) (order 3)", lambda: ndi.shift(xd, (0.5, -0.25, 0.75), output=out))]
        for name, fn in calls:
            t, _ = timeit(fn, 10)
            print(json.dumps({"shape": shape, "dtype": np.dtype(dt).name, "call": name, "us": round(t * 1e6, 1),
                              "of 8 TB/s": round(2 * x.nbytes / 8e12 / t, 3), "kernel": last_kernel()[4:48]}), flush=True)
        del xd, out
        ca.free_all_blocks()
