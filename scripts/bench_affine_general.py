"""affine_transform order 1 on 512^3 float32: rotations about axes that couple all three coordinates -- (1,1,1)/sqrt(3), the y axis
((z, x) plane coupled with ... no: about y is a (z, x) rotation, streamable), (1,0,1)/sqrt(2) -- by a sweep of angles: which kernel, how fast
-> profiles/r4_affine_general.txt"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import last_kernel, _lib
lib = _lib.load()
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
    for deg in ((2, 5, 7, 10, 15, 20, 25, 30, 45) if "--quick" not in sys.argv else (7, 15, 30)):
        M = rot(axis, deg)
        off = ctr - M @ ctr + np.array([0.5, -1.25, 2.0])
        row = {"axis": axis, "deg": deg}
        for kib, name in ((36, "r4 rule (36 KiB boxes, 64 / 32-wide tiles)"), (0, "r5")):
            # (r5: cube tiles and 64 KiB boxes; the r4 rule is emulated by the budget alone -- the cube then rarely wins)
            lib.mi_debug_set_affine_box_kib(kib)
            s_, f = timeit(lambda: ndi.affine_transform(xd, M, off, order=1, mode="constant", output=out), 10)
            row[name + " us"] = round(s_ * 1e6, 1)
            row[name + " frac"] = round(8 * n ** 3 / s_ / 8e12, 3)
            row[name + " kernel"] = last_kernel()[4:100]
        lib.mi_debug_set_affine_box_kib(0)
        lib.mi_debug_set_interp_c1(5)          # the L1-gather kernel, for reference
        s_, f = timeit(lambda: ndi.affine_transform(xd, M, off, order=1, mode="constant", output=out), 10)
        lib.mi_debug_set_interp_c1(1)
        row["L1 gathers us"] = round(s_ * 1e6, 1)
        print(json.dumps(row), flush=True)
