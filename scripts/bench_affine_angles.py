"""affine_transform order 1 on 512^3 float32: rotations by a sweep of angles in the (y, x) plane (axis 0 streams) and the (z, x)
plane (axis 1 streams), which kernel, how fast; the gather / box kernels as comparators -> profiles/r4_affine_angles.txt"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
from bench_configs import timeit
lib = _lib.load()
n = 512
SC = float(os.environ.get("AFFINE_SCALE", "1.02"))        # step along the streamed axis (<= 1: three ring slots)
x = fs.volume_f32((n,) * 3); xd = ca.asarray(x); out = ca.empty(x.shape, np.float32)
ctr = np.array([(n - 1) / 2.0] * 3)
for plane in ("yx", "zx"):
    for deg in (0, 3, 7, 15, 30, 45, 60, 75, 90, 120, 180):
        a = np.deg2rad(deg); c, s = np.cos(a), np.sin(a)
        M = np.array([[SC, 0, 0], [0, c, -s], [0, s, c]]) if plane == "yx" else np.array([[c, 0, -s], [0, SC, 0], [s, 0, c]])
        off = ctr - M @ ctr + np.array([0.5, -1.25, 2.0])
        row = {"plane": plane, "deg": deg}
        for knob, name in ((1, "auto"), (0, "without the streaming kernel")):
            lib.mi_debug_set_affine_zstream(knob)
            s_, f = timeit(lambda: ndi.affine_transform(xd, M, off, order=1, mode="constant", output=out), 10)
            row[name] = [round(s_ * 1e6, 1), round(8 * n ** 3 / s_ / 8e12, 3), last_kernel()[4:30] + last_kernel()[last_kernel().find("streams along it") + 17:last_kernel().find("streams along it") + 50] if "zstream" in last_kernel() else last_kernel()[4:34]]
        lib.mi_debug_set_affine_zstream(1)
        print(json.dumps(row), flush=True)
