"""affine_transform order 1 on 512^3 float32: rotations about axes that couple all three coordinates -- (1,1,1)/sqrt(3), the y axis
((z, x) plane coupled with ... no: about y is a (z, x) rotation, streamable), (1,0,1)/sqrt(2) -- by a sweep of angles: which kernel, how fast
-> profiles/r4_affine_general.txt"""
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
ctr = np.array([(n - 1) / 2.0] * 3)
def rot(axis, deg):
    a = np.deg2rad(deg); u = np.asarray(axis, float); u /= np.linalg.norm(u)
    K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)
for axis in ((1, 1, 1), (1, 0, 1), (0, 1, 1), (1, 1, 0)):
    for deg in (1, 2, 3, 5, 7, 10, 15, 20, 30, 45):
        M = rot(axis, deg)
        off = ctr - M @ ctr + np.array([0.5, -1.25, 2.0])
        s_, f = timeit(lambda: ndi.affine_transform(xd, M, off, order=1, mode="constant", output=out), 10)
        print(json.dumps({"axis": axis, "deg": deg, "us": round(s_ * 1e6, 1), "frac": round(8 * n ** 3 / s_ / 8e12, 3), "kernel": last_kernel()[4:80]}), flush=True)
