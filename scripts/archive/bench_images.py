"""2-D images (the shapes skimage callers pass): uniform 5, gaussian sigma 1 / 2, float32 / uint8 grey erosion,
median 3 -- time per call and algorithmic GB/s (read once + write once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi
from cupyimg_amd import _lib

if "IMG2D" in os.environ:
    _lib.load().mi_debug_set_sep3d_image2d(int(os.environ["IMG2D"]))
ONLY = os.environ.get("ONLY", "").split(",") if os.environ.get("ONLY") else None
if "SWZ" in os.environ:
    _lib.load().mi_debug_set_xcd_swizzle(int(os.environ["SWZ"]))
if "MINCHUNK" in os.environ:
    _lib.load().mi_debug_set_stream_min_chunk(int(os.environ["MINCHUNK"]))


def timeit(fn, reps=10):
    for _ in range(3): fn()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize()
    return e0.elapsed_ms(e1) / reps * 1e3


shapes = [(1024, 1024), (2048, 2048), (4096, 4096), (8192, 8192), (16384, 16384), (3000, 4000), (1080, 1920)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for shape in shapes:
    x = ca.asarray(np.random.default_rng(0).standard_normal(shape, dtype=np.float32))
    u = ca.asarray(np.random.default_rng(1).integers(0, 256, size=shape, dtype=np.uint8))
    o = ca.empty(shape, np.float32); uo = ca.empty(shape, np.uint8)
    n = float(np.prod(shape))
    print("shape %s" % (shape,), flush=True)
    for name, fn, bpv in [("uniform5", lambda: ndi.uniform_filter(x, size=5, output=o), 8),
                          ("uniform3", lambda: ndi.uniform_filter(x, size=3, output=o), 8),
                          ("uniform7", lambda: ndi.uniform_filter(x, size=7, output=o), 8),
                          ("gauss1", lambda: ndi.gaussian_filter(x, 1.0, output=o), 8),
                          ("gauss2", lambda: ndi.gaussian_filter(x, 2.0, output=o), 8),
                          ("gauss4", lambda: ndi.gaussian_filter(x, 4.0, output=o), 8),
                          ("erode5 f32", lambda: ndi.grey_erosion(x, size=5, output=o), 8),
                          ("erode7 u8", lambda: ndi.grey_erosion(u, size=7, output=uo), 2),
                          ("erode3 u8", lambda: ndi.grey_erosion(u, size=3, output=uo), 2),
                          ("median3 f32", lambda: ndi.median_filter(x, size=3, output=o), 8),
                          ("median3 u8", lambda: ndi.median_filter(u, size=3, output=uo), 2),
                          ("sobel f32", lambda: ndi.sobel(x, output=o), 8),
                          ("laplace f32", lambda: ndi.laplace(x, output=o), 8),
                          ("corr3x3 f32", lambda: ndi.correlate(x, np.ones((3, 3), np.float32), output=o), 8)]:
        if ONLY and not any(name.startswith(o) for o in ONLY):
            continue
        try:
            t = timeit(fn)
            print("   %-12s %8.1f us %6.0f GB/s  %4.1f %%" % (name, t, bpv * n / t / 1e3, bpv * n / t / 1e3 / 80.0), flush=True)
        except Exception as exc:
            print("   %-12s FAILED %s %s" % (name, type(exc).__name__, exc), flush=True)
    x = u = o = uo = None
    ca.free_all_blocks()
