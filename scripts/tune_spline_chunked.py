import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi
from cupyimg_amd import _lib
lib = _lib.load()
rng = np.random.default_rng(0)
def timeit(fn, reps=5):
    for _ in range(2): fn()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize()
    return e0.elapsed_ms(e1) / reps * 1e3
for shape in [(2048, 2048), (4096, 4096), (8192, 8192), (64, 512, 512)]:
    x = rng.standard_normal(shape).astype(np.float32); xd = ca.asarray(x)
    xd64 = ca.asarray(x.astype(np.float64))
    for thr in (65536, 262144, 1048576):
        for mc in (0, 64, 256):
            lib.mi_debug_set_spline_threads(thr); lib.mi_debug_set_spline_chunk(mc)
            t1 = timeit(lambda: ndi.spline_filter(xd64, order=3))
            t0 = timeit(lambda: ndi.spline_filter1d(xd64, order=3, axis=0))
            tl = timeit(lambda: ndi.spline_filter1d(xd64, order=3, axis=len(shape) - 1))
            print(shape, "threads", thr, "minchunk", mc, "spline_filter f64 %.0f us  axis0 %.0f us  last axis %.0f us" % (t1, t0, tl), flush=True)
