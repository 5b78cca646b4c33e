"""Typical MRI volume shapes: uniform 5, gaussian sigma 2, float32 grey erosion 5, uint8 grey erosion 5."""
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

for shape in [(180, 256, 256), (200, 320, 320), (160, 384, 384), (256, 256, 256), (192, 448, 448), (120, 512, 640)]:
    x = ca.asarray(np.random.default_rng(0).standard_normal(shape, dtype=np.float32))
    u = ca.asarray(np.random.default_rng(1).integers(0, 256, size=shape, dtype=np.uint8))
    o = ca.empty(shape, np.float32); uo = ca.empty(shape, np.uint8)
    n = float(np.prod(shape))
    row = "shape %-16s" % (shape,)
    for name, fn, bpv in [("uniform5", lambda: ndi.uniform_filter(x, size=5, output=o), 8), ("gauss2", lambda: ndi.gaussian_filter(x, 2.0, output=o), 8),
                          ("erode5 f32", lambda: ndi.grey_erosion(x, size=5, output=o), 8), ("erode5 u8", lambda: ndi.grey_erosion(u, size=5, output=uo), 2)]:
        try:
            t = timeit(fn)
            row += "  %s %6.1f us %4.0f GB/s" % (name, t, bpv * n / t / 1e3)
        except Exception as exc:
            row += "  %s FAILED %s" % (name, type(exc).__name__)
    print(row, flush=True)
    x = u = o = uo = None
    ca.free_all_blocks()
