"""Spline family: prefilter and order-3 interpolation, float32 volumes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi

def timeit(fn, reps=5):
    for _ in range(2): fn()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize()
    return e0.elapsed_ms(e1) / reps

rng = np.random.default_rng(0)
th = np.deg2rad(7.0)
M = np.diag([1.02, 1, 1]) @ np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]])
for n in (256, 512):
    x = ca.asarray(rng.standard_normal((n, n, n), dtype=np.float32))
    c = (n - 1) / 2
    off = np.array([c, c, c]) - M @ np.array([c, c, c]) + np.array([0.5, -1.25, 2.0])
    for axis in (0, 1, 2):
        t = timeit(lambda: ndi.spline_filter1d(x, 3, axis=axis, output=np.float64), 3)
        print("spline_filter1d order 3 axis %d  %d^3 f32->f64: %8.3f ms" % (axis, n, t), flush=True)
    t = timeit(lambda: ndi.spline_filter(x, 3, output=np.float64), 3)
    print("spline_filter   order 3          %d^3 f32->f64: %8.3f ms" % (n, t), flush=True)
    for axis in (0, 1, 2):
        t = timeit(lambda: ndi.spline_filter1d(x, 3, axis=axis, output=np.float32, allow_float32=True), 3)
        print("spline_filter1d order 3 axis %d  %d^3 f32 coefficients: %8.3f ms" % (axis, n, t), flush=True)
    coef = ndi.spline_filter(x, 3, output=np.float64)
    for order in (2, 3, 5):
        cf = ndi.spline_filter(x, order, output=np.float64)
        t = timeit(lambda: ndi.affine_transform(cf, M, offset=off, order=order, prefilter=False, output=np.float32), 3)
        print("affine_transform order %d prefilter=False %d^3: %8.3f ms  (%7.0f Mvox/s)" % (order, n, t, n ** 3 / t / 1e3), flush=True)
        cf = None
    t = timeit(lambda: ndi.affine_transform(x, M, offset=off, order=3), 3)
    print("affine_transform order 3 (with prefilter)  %d^3: %8.3f ms  (%7.0f Mvox/s)" % (n, t, n ** 3 / t / 1e3), flush=True)
    t = timeit(lambda: ndi.zoom(x, 1.25, order=3), 3)
    print("zoom 1.25 order 3                          %d^3: %8.3f ms" % (n, t), flush=True)
    x = coef = None
    ca.free_all_blocks()
x2 = ca.asarray(rng.standard_normal((4096, 4096), dtype=np.float32))
t = timeit(lambda: ndi.rotate(x2, 13.0, order=3, reshape=False), 3)
print("rotate 13 deg order 3 4096^2: %8.3f ms (%7.0f Mpix/s)" % (t, 4096 ** 2 / t / 1e3))
t = timeit(lambda: ndi.spline_filter(x2, 3, output=np.float64), 3)
print("spline_filter order 3 4096^2: %8.3f ms" % t)
