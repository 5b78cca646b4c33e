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
M = rot((1, 1, 1), 5.0); off = ctr - M @ ctr + np.array([0.5, -1.25, 2.0])
for knob in (0, 1, 3, 5, 7, 9, 15):
    lib.mi_debug_set_cubic_box(knob)
    t, _ = timeit(lambda: ndi.affine_transform(xd, M, off, order=3, prefilter=False, output=out), 4)
    print(knob, round(t * 1e6, 1), last_kernel()[4:60], flush=True)
lib.mi_debug_set_cubic_box(0)
