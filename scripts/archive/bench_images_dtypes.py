"""2-D images in the other dtypes skimage pipelines use (float64 from img_as_float, uint8 / uint16 raw data, bool masks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi

def timeit(fn, reps=10):
    for _ in range(3): fn()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize()
    return e0.elapsed_ms(e1) / reps * 1e3

shapes = [(4096, 4096), (8192, 8192)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for shape in shapes:
    rng = np.random.default_rng(0)
    n = float(np.prod(shape))
    arrs = {"f64": rng.standard_normal(shape), "f32": rng.standard_normal(shape, dtype=np.float32),
            "u8": rng.integers(0, 256, size=shape, dtype=np.uint8), "u16": rng.integers(0, 65536, size=shape, dtype=np.uint16),
            "bool": rng.random(shape) > 0.3}
    print("shape %s" % (shape,), flush=True)
    for dt, h in arrs.items():
        x = ca.asarray(h)
        o = ca.empty(shape, h.dtype)
        isz = h.dtype.itemsize
        ops = [("uniform5", lambda: ndi.uniform_filter(x, size=5, output=o)), ("gauss2", lambda: ndi.gaussian_filter(x, 2.0, output=o)),
               ("erode5", lambda: ndi.grey_erosion(x, size=5, output=o)), ("median3", lambda: ndi.median_filter(x, size=3, output=o)),
               ("sobel", lambda: ndi.sobel(x, output=o)), ("corr3x3", lambda: ndi.correlate(x, np.ones((3, 3)) / 9.0, output=o))]
        if dt == "bool":
            ops = [("binary_erosion", lambda: ndi.binary_erosion(x, output=o)), ("binary_erosion it3", lambda: ndi.binary_erosion(x, iterations=3, output=o)),
                   ("binary_dilation 3x3", lambda: ndi.binary_dilation(x, structure=np.ones((3, 3), bool), output=o)),
                   ("binary_opening", lambda: ndi.binary_opening(x, output=o)), ("binary_fill_holes", None)]
        if dt in ("f32", "f64"):
            of = ca.empty(shape, h.dtype)
            ops += [("affine o1", lambda: ndi.affine_transform(x, np.array([[0.98, 0.05], [-0.05, 0.98]]), offset=(3.0, -2.0), order=1, output=of)),
                    ("affine o3", lambda: ndi.affine_transform(x, np.array([[0.98, 0.05], [-0.05, 0.98]]), offset=(3.0, -2.0), order=3, output=of)),
                    ("zoom1.5 o1", lambda: ndi.zoom(x, 1.5, order=1)), ("rotate30 o1", lambda: ndi.rotate(x, 30.0, order=1, reshape=False, output=of)),
                    ("shift o1", lambda: ndi.shift(x, (2.5, -1.25), order=1, output=of))]
        for name, fn in ops:
            if fn is None:
                continue
            try:
                t = timeit(fn, reps=5)
                print("   %-5s %-20s %9.1f us %6.0f GB/s  %4.1f %%" % (dt, name, t, 2 * isz * n / t / 1e3, 2 * isz * n / t / 1e3 / 80.0), flush=True)
            except Exception as exc:
                print("   %-5s %-20s FAILED %s %s" % (dt, name, type(exc).__name__, str(exc)[:80]), flush=True)
        x = o = None
        ca.free_all_blocks()
