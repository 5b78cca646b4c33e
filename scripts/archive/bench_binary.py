"""binary erosion / dilation throughput (LDS-tiled binary3d.hip vs generic kernel)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi

lib = _lib.load()
lib.mi_debug_set_binary_tiled.argtypes = [ctypes.c_int]

def timeit(fn, reps=5):
    for _ in range(2): fn()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize()
    return e0.elapsed_ms(e1) / reps

rng = np.random.default_rng(0)
for n in (256, 512, 1024):
    b = ca.asarray(rng.random((n, n, n)) > 0.4)
    bo = ca.empty(b.shape, np.bool_)
    for name, st in [("cross (7 taps)", None), ("3x3x3 full", np.ones((3, 3, 3), bool)), ("5x5x5 full", np.ones((5, 5, 5), bool))]:
        res = []
        for en in (1, 0):
            lib.mi_debug_set_binary_tiled(en)
            res.append(timeit(lambda: ndi.binary_erosion(b, st, output=bo), 3))
        lib.mi_debug_set_binary_tiled(1)
        print("binary_erosion %-15s bool %4d^3  tiled %8.3f ms (%6.0f GB/s alg = %4.1f%% of 8 TB/s)  generic %8.3f ms" % (
            name, n, res[0], 2 * n ** 3 / res[0] / 1e6, 2 * n ** 3 / res[0] / 1e6 / 80, res[1]), flush=True)
    b = bo = None
    ca.free_all_blocks()
