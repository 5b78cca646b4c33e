"""Timings of the generic kernels on large inputs (where the time goes outside the BASELINE configs)."""
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
n = 256
x = ca.asarray(rng.standard_normal((n, n, n), dtype=np.float32))
o = ca.empty(x.shape, np.float32)
vox = n ** 3 / 1e6
def rep(name, t): print("%-46s %9.3f ms  %9.0f Mvox/s" % (name, t, vox / t * 1e3))
w3 = rng.standard_normal((3, 3, 3)); w5 = rng.standard_normal((5, 5, 5))
rep("correlate 3x3x3 f32 256^3", timeit(lambda: ndi.correlate(x, w3, output=o)))
rep("correlate 5x5x5 f32 256^3", timeit(lambda: ndi.correlate(x, w5, output=o)))
rep("correlate1d 5 taps axis0 f32 (generic, f64 acc)", timeit(lambda: ndi.correlate1d(x, [1, 2, 3, 2, 1.], axis=0, output=o, dtype_mode="ndimage")))
rep("correlate1d 5 taps axis2 f32 (generic, f32 acc)", timeit(lambda: ndi.correlate1d(x, [1, 2, 3, 2, 1.], axis=2, output=o)))
fp = rng.random((3, 3, 3)) > 0.3
rep("maximum_filter footprint 3x3x3 f32", timeit(lambda: ndi.maximum_filter(x, footprint=fp, output=o)))
b = ca.asarray(rng.random((n, n, n)) > 0.4)
bo = ca.empty(b.shape, np.bool_)
rep("binary_erosion default struct bool 256^3", timeit(lambda: ndi.binary_erosion(b, output=bo)))
rep("binary_dilation 3 iterations", timeit(lambda: ndi.binary_dilation(b, iterations=3, output=bo)))
u = ca.asarray(rng.integers(0, 255, size=(n, n, n), dtype=np.uint8).astype(np.int16))
uo = ca.empty(u.shape, np.int16)
rep("uniform_filter size 5 int16 (generic passes)", timeit(lambda: ndi.uniform_filter(u, 5, output=uo)))
x2 = ca.asarray(rng.standard_normal((4096, 4096), dtype=np.float32)); o2 = ca.empty(x2.shape, np.float32)
vox = 4096 * 4096 / 1e6
rep("gaussian_filter sigma 2, 4096^2 f32 (2-D)", timeit(lambda: ndi.gaussian_filter(x2, 2.0, output=o2)))
rep("uniform_filter 5, 4096^2 f32 (2-D)", timeit(lambda: ndi.uniform_filter(x2, 5, output=o2)))
