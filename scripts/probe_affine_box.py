"""r5 probe: LDS-staged general affine kernel -- tile shape x box budget per angle (512^3 float32, rotation about (1, 1, 1)).  usage: python scripts/probe_affine_box.py"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import last_kernel, _lib
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
from bench_configs import timeit
lib = _lib.load()
n = 512
x = fs.volume_f32((n,) * 3); xd = ca.asarray(x); out = ca.empty(x.shape, np.float32)
ctr = np.array([(n - 1) / 2.0] * 3)
def rot(axis, deg):
    a = np.deg2rad(deg); u = np.asarray(axis, float); u /= np.linalg.norm(u)
    K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)
for axis in ((1, 1, 1), (1, 1, 0)):
    for deg in (10, 20, 30, 45):
        M = rot(axis, deg); off = ctr - M @ ctr + np.array([0.5, -1.25, 2.0])
        row = {"axis": axis, "deg": deg}
        for shape, nm in ((1, "64x8x8"), (2, "32x16x8"), (3, "16x16x16"), (4, "16x32x8")):
            lib.mi_debug_set_affine_box_kib(1000 * shape + 128)
            s_, _ = timeit(lambda: ndi.affine_transform(xd, M, off, order=1, mode="constant", output=out), 10)
            k = last_kernel()
            row[nm] = (round(s_ * 1e6, 1), k[k.find("tile"):k.find("box")].split(",")[1].strip() if "lds_kernel" in k else "gathers")
        lib.mi_debug_set_affine_box_kib(0)
        print(json.dumps(row), flush=True)
