"""More 2-D cases: dense correlate 5x5 / 7x7, footprint (disk) erosion, composites, skimage facade calls."""
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
    return e0.elapsed_ms(e1) / reps * 1e3

def disk(r):
    y, x = np.mgrid[-r:r + 1, -r:r + 1]
    return (x * x + y * y) <= r * r

shape = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (8192, 8192)
rng = np.random.default_rng(0)
n = float(np.prod(shape))
for dt in ("float32", "uint8", "float64"):
    h = rng.standard_normal(shape).astype(dt) if dt != "uint8" else rng.integers(0, 256, size=shape, dtype=np.uint8)
    x = ca.asarray(h); o = ca.empty(shape, h.dtype)
    isz = h.dtype.itemsize
    k5 = rng.standard_normal((5, 5)); k7 = rng.standard_normal((7, 7))
    ops = [("correlate 5x5", lambda: ndi.correlate(x, k5, output=o)), ("correlate 7x7", lambda: ndi.correlate(x, k7, output=o)),
           ("erode disk1", lambda: ndi.grey_erosion(x, footprint=disk(1), output=o)), ("erode disk2", lambda: ndi.grey_erosion(x, footprint=disk(2), output=o)),
           ("erode disk3", lambda: ndi.grey_erosion(x, footprint=disk(3), output=o)), ("median disk2", lambda: ndi.median_filter(x, footprint=disk(2), output=o)),
           ("median 5x5", lambda: ndi.median_filter(x, size=5, output=o)),
           ("gaussian_laplace 1.5", lambda: ndi.gaussian_laplace(x, 1.5, output=o)), ("gauss_grad_mag 1.5", lambda: ndi.gaussian_gradient_magnitude(x, 1.5, output=o)),
           ("prewitt", lambda: ndi.prewitt(x, output=o)), ("white_tophat 5", lambda: ndi.white_tophat(x, size=5, output=o)),
           ("morph_gradient 3", lambda: ndi.morphological_gradient(x, size=3, output=o)), ("grey_opening 5", lambda: ndi.grey_opening(x, size=5, output=o))]
    for name, fn in ops:
        try:
            t = timeit(fn)
            print("   %-8s %-22s %9.1f us %6.0f GB/s  %4.1f %%" % (dt, name, t, 2 * isz * n / t / 1e3, 2 * isz * n / t / 1e3 / 80.0), flush=True)
        except Exception as exc:
            print("   %-8s %-22s FAILED %s %s" % (dt, name, type(exc).__name__, str(exc)[:80]), flush=True)
    x = o = None
    ca.free_all_blocks()
