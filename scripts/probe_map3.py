import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
from bench_configs import timeit
n = 512
x = fs.volume_f32((n,) * 3); xd = ca.asarray(x); out = ca.empty(x.shape, np.float32)
cd = ca.asarray(fs.affine_coords_f32(n))
for order, pre in ((1, True), (3, True), (3, False)):  # (order 3: cubic3_mapbox_kernel since r5)
    t, _ = timeit(lambda: ndi.map_coordinates(xd, cd, order=order, prefilter=pre, output=out), 4)
    print(json.dumps({"call": "map_coordinates order %d prefilter=%s" % (order, pre), "us": round(t * 1e6, 1), "kernel": last_kernel()[4:60]}), flush=True)
for axes in ((1, 0), (2, 1), (2, 0)):
    t, _ = timeit(lambda: ndi.rotate(xd, 7.0, axes=axes, reshape=False, output=out), 4)
    print(json.dumps({"call": "rotate(v, 7, axes=%s)" % (axes,), "us": round(t * 1e6, 1), "kernel": last_kernel()[4:60]}), flush=True)
