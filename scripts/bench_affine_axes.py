"""order-1 affine on 512^3: which kernel and how fast for rotations in each coordinate plane + a general one"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit
n = 512
xd = ca.empty((n,) * 3, np.float32); xd.fill(1.0)
out = ca.empty((n,) * 3, np.float32)
def rot(i, j, deg):
    a = np.deg2rad(deg); R = np.eye(3); R[i, i] = np.cos(a); R[i, j] = -np.sin(a); R[j, i] = np.sin(a); R[j, j] = np.cos(a); return R
ctr = (n - 1) / 2.0
cases = {"rot(y,x) 7deg [axis 0 decoupled]": rot(1, 2, 7), "rot(z,x) 7deg [axis 1 decoupled]": rot(0, 2, 7), "rot(z,y) 7deg [axis 2 decoupled: scipy rotate default]": rot(0, 1, 7),
         "rot(z,y) 30deg": rot(0, 1, 30), "rot(y,x) 30deg": rot(1, 2, 30), "general: 7deg about each axis": rot(0, 1, 7) @ rot(0, 2, 7) @ rot(1, 2, 7), "zoom 1.1 isotropic (diag)": np.eye(3) / 1.1}
for name, M in cases.items():
    off = ctr - M @ np.array([ctr] * 3)
    s, f = timeit(lambda: ndi.affine_transform(xd, M, off, order=1, mode="constant", output=out), 30)
    print(json.dumps({"case": name, "us": round(s * 1e6, 1), "frac": round(8 * n**3 / s / 8e12, 3), "kernel": last_kernel()[:70]}), flush=True)
