"""Order-3 affine transforms of a 512^3 float32 volume whose matrix leaves axis 0 to itself (`rotate(volume, a, axes=(1, 2))`
with scipy's default order) or the x axis (`rotate(volume, a)`, cubic3_rowblend_kernel, at the end): the z-streaming kernel (cubic3_zstream_kernel) against the gather kernel it replaces
(cubic3_f32_kernel, debug knob 0), with and without the prefilter, per angle and for the boundary modes that pad the
coefficient array.  One JSON line per case.  usage: python scripts/bench_cubic_affine.py [--quick]"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit

lib = _lib.load()
rng = np.random.default_rng(0)
n = 512
x = rng.standard_normal((n,) * 3).astype(np.float32); xd = ca.asarray(x); out = ca.empty(x.shape, np.float32)
quick = "--quick" in sys.argv
if "--counters" in sys.argv:
    # for scripts/pmc_configs.sh: the default kernels only, 7 degrees in each of the three planes, with the prefilter
    ctr = np.array([(n - 1) / 2] * 3); a = np.deg2rad(7.0); c, s = np.cos(a), np.sin(a)
    for M in (np.array([[1.0, 0, 0], [0, c, -s], [0, s, c]]), np.array([[c, 0, -s], [0, 1.0, 0], [s, 0, c]]), np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])):
        for _ in range(6):
            ndi.affine_transform(xd, M, ctr - M @ ctr, order=3, output=out)
    out.get()
    sys.exit(0)
angles = (7.0, 30.0) if quick else (0.5, 2.0, 4.0, 7.0, 10.0, 15.0, 30.0, 45.0, 60.0, 75.0, 80.0, 85.0, 88.0, 90.0, 135.0, 180.0, 187.0)
for deg in angles:
    a = np.deg2rad(deg); M = np.array([[1.0, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    ctr = np.array([(n - 1) / 2] * 3); off = ctr - M @ ctr
    row = {"deg": deg}
    for knob, name in ((0, "gather"), (1, "default")):
        lib.mi_debug_set_cubic_zstream(knob)
        s_, _ = timeit(lambda: ndi.affine_transform(xd, M, off, order=3, prefilter=False, output=out), 5)
        row["%s us (prefilter=False)" % name] = round(s_ * 1e6, 1)
        row["%s kernel" % name] = last_kernel()[4:26]
        s_, _ = timeit(lambda: ndi.affine_transform(xd, M, off, order=3, output=out), 4)
        row["%s us (with prefilter)" % name] = round(s_ * 1e6, 1)
    lib.mi_debug_set_cubic_zstream(1)
    # algorithmic bytes of the interpolation: one read of the coefficients + one write of the output
    row["hbm roofline frac (prefilter=False)"] = round(2 * x.nbytes / 8e12 / (row["default us (prefilter=False)"] * 1e-6), 3)
    print(json.dumps(row), flush=True)
a = np.deg2rad(7.0); M = np.array([[1.0, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
ctr = np.array([(n - 1) / 2] * 3); off = ctr - M @ ctr
for mode in ("mirror", "nearest", "grid-wrap"):
    row = {"deg": 7.0, "mode": mode}
    for knob, name in ((0, "gather"), (1, "default")):
        lib.mi_debug_set_cubic_zstream(knob)
        s_, _ = timeit(lambda: ndi.affine_transform(xd, M, off, order=3, mode=mode, output=out), 4)
        row["%s us (with prefilter)" % name] = round(s_ * 1e6, 1)
        row["%s kernel" % name] = last_kernel()[4:26]
    lib.mi_debug_set_cubic_zstream(1)
    print(json.dumps(row), flush=True)
# axis 1 to itself: rotations in the (z, x) plane (cubic3_zstream_kernel<1>, streams along y)
for deg in ((7.0,) if quick else (7.0, 30.0, 80.0)):
    a = np.deg2rad(deg); M = np.array([[np.cos(a), 0, -np.sin(a)], [0, 1.0, 0], [np.sin(a), 0, np.cos(a)]])
    off = ctr - M @ ctr
    row = {"plane": "(z, x)", "deg": deg}
    for knob, name in ((0, "gather"), (1, "default")):
        lib.mi_debug_set_cubic_zstream(knob)
        s_, _ = timeit(lambda: ndi.affine_transform(xd, M, off, order=3, prefilter=False, output=out), 5)
        row["%s us (prefilter=False)" % name] = round(s_ * 1e6, 1)
        row["%s kernel" % name] = last_kernel()[4:29]
        s_, _ = timeit(lambda: ndi.affine_transform(xd, M, off, order=3, output=out), 4)
        row["%s us (with prefilter)" % name] = round(s_ * 1e6, 1)
    lib.mi_debug_set_cubic_zstream(1)
    row["hbm roofline frac (prefilter=False)"] = round(2 * x.nbytes / 8e12 / (row["default us (prefilter=False)"] * 1e-6), 3)
    print(json.dumps(row), flush=True)
# the x axis to itself: rotations in the (z, y) plane, SciPy's default axes for `rotate` (cubic3_rowblend_kernel)
for deg in ((7.0,) if quick else (7.0, 30.0, 90.0)):
    a = np.deg2rad(deg); M = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    off = ctr - M @ ctr
    row = {"plane": "(z, y)", "deg": deg}
    for knob, name in ((0, "gather"), (1, "default")):
        lib.mi_debug_set_cubic_rowblend(knob)
        s_, _ = timeit(lambda: ndi.affine_transform(xd, M, off, order=3, prefilter=False, output=out), 5)
        row["%s us (prefilter=False)" % name] = round(s_ * 1e6, 1)
        row["%s kernel" % name] = last_kernel()[4:26]
        s_, _ = timeit(lambda: ndi.affine_transform(xd, M, off, order=3, output=out), 4)
        row["%s us (with prefilter)" % name] = round(s_ * 1e6, 1)
    lib.mi_debug_set_cubic_rowblend(1)
    row["hbm roofline frac (prefilter=False)"] = round(2 * x.nbytes / 8e12 / (row["default us (prefilter=False)"] * 1e-6), 3)
    print(json.dumps(row), flush=True)
for mode in ("constant", "mirror", "nearest"):
    s_, _ = timeit(lambda: ndi.rotate(xd, 7.0, reshape=False, mode=mode, output=out), 4)
    print(json.dumps({"what": "rotate(volume, 7, reshape=False, mode=%r): SciPy's default axes (1, 0) and order 3" % mode, "us": round(s_ * 1e6, 1), "kernel": last_kernel()[4:26]}), flush=True)
s_, _ = timeit(lambda: ndi.rotate(xd, 7.0, axes=(1, 2), reshape=False, output=out), 4)
print(json.dumps({"what": "rotate(volume, 7, axes=(1, 2), reshape=False), scipy's default order 3", "us": round(s_ * 1e6, 1), "kernel": last_kernel()[4:26]}), flush=True)
