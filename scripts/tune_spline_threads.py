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
for shape in [(2048, 2048), (4096, 4096), (8192, 8192), (3000, 4000)]:
    xd64 = ca.asarray(rng.standard_normal(shape)); xd32 = ca.asarray(rng.standard_normal(shape).astype(np.float32))
    row = "%-14s" % (shape,)
    for thr in (24576, 32768, 49152, 65536, 98304, 131072):
        lib.mi_debug_set_spline_threads(thr)
        t1 = timeit(lambda: ndi.spline_filter(xd64, order=3))
        t2 = timeit(lambda: ndi.rotate(xd32, 13.0, order=3, reshape=False))
        row += "  T=%d: %.0f / %.0f" % (thr, t1, t2)
    print(row, flush=True)
