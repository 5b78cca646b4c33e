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
for shape in [(4096, 4096), (8192, 8192)]:
    xd64 = ca.asarray(rng.standard_normal(shape))
    o = ca.empty(shape, np.float64)
    for chunk, rows in [(-1, 1), (-1, 2), (0, 1)]:
        lib.mi_debug_set_spline_chunk(chunk); lib.mi_debug_set_spline_rows(rows)
        def f1():
            o[...] = xd64
            lib_call = ndi.spline_filter1d(xd64, order=3, axis=1, output=o)
        t = timeit(f1)
        print(shape, "chunk", chunk, "rows", rows, "axis-1 pass incl. copy %.0f us" % t, flush=True)
