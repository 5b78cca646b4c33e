"""Kernels longer than 17 taps: x pass fused into the streamed pass (FUSED=33, default) or separate (FUSED=17)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi
from cupyimg_amd import _lib
lib = _lib.load()

def timeit(fn, reps=10):
    for _ in range(3): fn()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize()
    return e0.elapsed_ms(e1) / reps * 1e3

for shape in [(8192, 8192), (4096, 4096), (512, 512, 512), (256, 512, 512)]:
    x = ca.asarray(np.random.default_rng(0).standard_normal(shape, dtype=np.float32))
    o = ca.empty(shape, np.float32)
    n = float(np.prod(shape))
    for sigma in (2.25, 2.5, 3.0, 4.0):
        row = "shape %-16s sigma %.2f (%d taps)" % (shape, sigma, 2 * int(4 * sigma + 0.5) + 1)
        for fused in (17, 33):
            lib.mi_debug_set_stream_fused_max(fused)
            t = timeit(lambda: ndi.gaussian_filter(x, sigma, output=o))
            row += "   fused<=%d: %8.1f us %5.0f GB/s" % (fused, t, 8 * n / t / 1e3)
        print(row, flush=True)
    x = o = None
    ca.free_all_blocks()
