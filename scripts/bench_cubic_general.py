"""r5: order-3 affine_transform on 512^3 float32 with matrices that couple all three axes -- cubic3_box_kernel (LDS-staged box) against
the gather kernel (mi_debug_set_cubic_box(0)); prefilter=False times the interpolation kernel alone -> profiles/r5_cubic_general.txt"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit
lib = _lib.load()
n = 512
x = np.random.default_rng(0).standard_normal((n,) * 3).astype(np.float32)
xd = ca.asarray(x); out = ca.empty(x.shape, np.float32)
ctr = np.array([(n - 1) / 2.0] * 3)
def rot(axis, deg):
    a = np.deg2rad(deg); u = np.asarray(axis, float); u /= np.linalg.norm(u)
    K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)
for axis in ((1, 1, 1), (1, 0, 1), (0.3, 1, -0.5)):
    for deg in (1, 3, 5, 7, 10, 15, 20, 30, 45):
        M = rot(axis, deg); off = ctr - M @ ctr + np.array([0.5, -1.25, 2.0])
        row = {"axis": axis, "deg": deg}
        for knob, name in ((1, "box kernel"), (0, "gather")):
            lib.mi_debug_set_cubic_box(knob)
            t, _ = timeit(lambda: ndi.affine_transform(xd, M, off, order=3, prefilter=False, output=out), 4)
            row[name + " us"] = round(t * 1e6, 1)
            if knob: row["kernel"] = last_kernel()[4:24]; row["of 8 TB/s"] = round(2 * x.nbytes / 8e12 / t, 3)
        lib.mi_debug_set_cubic_box(1)
        t, _ = timeit(lambda: ndi.affine_transform(xd, M, off, order=3, output=out), 4)
        row["whole call with prefilter us"] = round(t * 1e6, 1)
        print(json.dumps(row), flush=True)
