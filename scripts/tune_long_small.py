import os, sys
sys.path.insert(0, "/root/repo")
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
for shape in [(128, 128, 128), (180, 256, 256), (256, 256, 256), (160, 384, 384), (192, 448, 448), (320, 512, 512), (512, 512, 512)]:
    x = ca.asarray(np.random.default_rng(0).standard_normal(shape, dtype=np.float32)); o = ca.empty(shape, np.float32)
    row = "%-16s" % (shape,)
    for sigma in (1.0, 1.5, 2.0):
        for hook in (0, 1):
            lib.mi_debug_set_sep3d_long(hook)
            t = timeit(lambda: ndi.gaussian_filter(x, sigma, output=o))
            row += "  s=%.1f %s %6.1f" % (sigma, "long" if hook == 0 else "strm", t)
    print(row, flush=True)
    for z in (0, 2, 4, 8, 16):
        lib.mi_debug_set_sep3d_long(0); lib.mi_debug_set_long_zchunks(z)
        t = timeit(lambda: ndi.gaussian_filter(x, 2.0, output=o))
        print("      sigma 2 long zchunks=%d: %.1f us" % (z, t), flush=True)
    lib.mi_debug_set_long_zchunks(0)
