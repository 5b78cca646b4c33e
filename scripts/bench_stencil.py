"""Dense n-D correlate (LDS-tiled stencil3d.hip vs the generic gather kernel)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi

lib = _lib.load()
lib.mi_debug_set_stencil.argtypes = [ctypes.c_int]
lib.mi_debug_set_stencil_scatter.argtypes = [ctypes.c_int]

def timeit(fn, reps=5):
    for _ in range(2): fn()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize()
    return e0.elapsed_ms(e1) / reps

rng = np.random.default_rng(0)
for n in (256, 512):
    x = ca.asarray(rng.standard_normal((n, n, n), dtype=np.float32))
    o = ca.empty(x.shape, np.float32)
    for wshape in [(3, 3, 3), (5, 5, 5), (7, 7, 7), (3, 3, 1), (1, 5, 5)]:
        w = rng.standard_normal(wshape)
        for dm in ("ndimage", "float", "ndimage/f32-weights"):
            if dm.endswith("f32-weights"):
                if wshape not in ((3, 3, 3), (5, 5, 5)):
                    continue
                w = w.astype(np.float32)                # float32-valued weights: exact products, v_fma_f64 (stencil3s.hip)
                dm = "ndimage"
            res = []
            lib.mi_debug_set_stencil_scatter(0)
            ring = timeit(lambda: ndi.correlate(x, w, output=o, dtype_mode=dm), 3)
            lib.mi_debug_set_stencil_scatter(1)
            for en in (1, 0):
                if en == 0 and (n == 512 and np.prod(wshape) > 27):
                    res.append(float("nan")); continue
                lib.mi_debug_set_stencil(en)
                res.append(timeit(lambda: ndi.correlate(x, w, output=o, dtype_mode=dm), 3))
            lib.mi_debug_set_stencil(1)
            t = res[0]
            print("correlate %s f32 %d^3 acc=%-7s %8.3f ms (%6.0f GB/s alg, %4.1f%% of 8 TB/s)   LDS-ring kernel %8.3f ms   generic %8.3f ms   %s" % (
                "x".join(map(str, wshape)), n, "f64" if dm == "ndimage" else "f32", t, 8 * n ** 3 / t / 1e6,
                8 * n ** 3 / t / 1e6 / 80, ring, res[1], ca.last_kernel()[:34]), flush=True)
    x = o = None
    ca.free_all_blocks()
x2 = ca.asarray(rng.standard_normal((8192, 8192), dtype=np.float32)); o2 = ca.empty(x2.shape, np.float32)
for wshape in [(3, 3), (5, 5), (7, 7)]:
    w = rng.standard_normal(wshape)
    t = timeit(lambda: ndi.correlate(x2, w, output=o2), 3)
    print("correlate %s f32 8192^2 tiled %8.3f ms (%6.0f GB/s alg)" % ("x".join(map(str, wshape)), t, 8 * 8192 ** 2 / t / 1e6))
