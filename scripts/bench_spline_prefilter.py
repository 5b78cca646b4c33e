"""r5: the B-spline prefilter on 512^3 (and two other shapes) -- every pass on its own and the public calls, one-sweep kernels
(csrc/spline_fast.hip, default) against the sequential ones (mi_debug_set_spline_fast(0)); then the default-order calls the
prefilter sits in front of: rotate with SciPy's default axes, affine_transform(order=3) with the BASELINE matrix, a general
three-axis matrix, zoom.  One JSON line per row -> profiles/r5_spline_prefilter.txt.   usage: python scripts/bench_spline_prefilter.py [--quick]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit
from helpers import fullsize as fs

lib = _lib.load()
quick = "--quick" in sys.argv
rng = np.random.default_rng(0)


def both(fn, reps=5):
    """(fast us, sequential us, kernel of the fast call)"""
    lib.mi_debug_set_spline_fast(1)
    t1, _ = timeit(fn, reps)
    k = last_kernel()[4:40]
    lib.mi_debug_set_spline_fast(0)
    try:
        t0, _ = timeit(fn, reps)
    finally:
        lib.mi_debug_set_spline_fast(1)
    return round(t1 * 1e6, 1), round(t0 * 1e6, 1), k


for shape in ((512, 512, 512),) if quick else ((512, 512, 512), (256, 256, 256), (181, 217, 184), (320, 640, 768)):
    x = rng.standard_normal(shape).astype(np.float32)
    xd = ca.asarray(x)
    nb = x.nbytes
    for dt in (np.float32, np.float64):
        out = ca.empty(shape, dt)
        src = xd if dt == np.float32 else xd.astype(np.float64)
        for axis in range(3):
            f, s, k = both(lambda: ndi.spline_filter1d(src, 3, axis=axis, output=out))
            alg = 2 * out.nbytes              # one read + one write of the coefficients
            print(json.dumps({"shape": shape, "coef": dt.__name__, "call": "spline_filter1d axis %d" % axis, "fast us": f, "sequential us": s,
                              "kernel": k, "of 8 TB/s": round(alg / 8e12 / (f * 1e-6), 3)}), flush=True)
        del src
        f, s, k = both(lambda: ndi.spline_filter(xd, 3, output=out))
        alg = nb + out.nbytes                 # read the samples once, write the coefficients once: what ONE fused launch would move
        print(json.dumps({"shape": shape, "coef": dt.__name__, "call": "spline_filter (float32 in)", "fast us": f, "sequential us": s,
                          "of 8 TB/s (one read + one write)": round(alg / 8e12 / (f * 1e-6), 3)}), flush=True)
        del out
    out = ca.empty(shape, np.float32)
    n = np.array(shape)
    ctr = (n - 1) / 2.0
    f, s, k = both(lambda: ndi.rotate(xd, 7.0, reshape=False, output=out))
    print(json.dumps({"shape": shape, "call": "rotate(v, 7, reshape=False) every default", "fast us": f, "sequential us": s, "kernel": k}), flush=True)
    M, off = fs.affine_case(shape[0]) if shape[0] == shape[1] == shape[2] else (None, None)
    if M is not None:
        f, s, k = both(lambda: ndi.affine_transform(xd, M, off, order=3, output=out))
        print(json.dumps({"shape": shape, "call": "affine_transform(order=3) BASELINE matrix", "fast us": f, "sequential us": s, "kernel": k}), flush=True)
        t, _ = timeit(lambda: ndi.affine_transform(xd, M, off, order=3, prefilter=False, output=out), 5)
        print(json.dumps({"shape": shape, "call": "  the same, prefilter=False (interpolation kernel alone)", "us": round(t * 1e6, 1), "kernel": last_kernel()[4:40],
                          "of 8 TB/s": round(2 * nb / 8e12 / t, 3)}), flush=True)
    a, b = np.deg2rad(9.0), np.deg2rad(-14.0)
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    Mg = Rz @ Rx
    f, s, k = both(lambda: ndi.affine_transform(xd, Mg, ctr - Mg @ ctr, order=3, output=out))
    print(json.dumps({"shape": shape, "call": "affine_transform(order=3) three axes coupled (9 and -14 degrees)", "fast us": f, "sequential us": s, "kernel": k}), flush=True)
    f, s, k = both(lambda: ndi.zoom(xd, 1.0, output=out) if False else ndi.shift(xd, (0.5, -0.25, 0.75), output=out))
    print(json.dumps({"shape": shape, "call": "shift(v, (0.5, -0.25, 0.75)) every default", "fast us": f, "sequential us": s, "kernel": k}), flush=True)
    del out, xd
    ca.free_all_blocks()
