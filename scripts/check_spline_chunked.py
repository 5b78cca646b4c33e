import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, scipy.ndimage as sndi
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
for shape in [(300, 520), (1024, 1024), (2048, 4096), (40, 200, 300)]:
    x = rng.standard_normal(shape)
    xd = ca.asarray(x)
    for order in (2, 3):
        for mode in ("mirror", "reflect", "nearest", "constant"):
            ref = sndi.spline_filter(x, order=order, mode=mode)
            for hook in (-1, 0, 16, 128):
                lib.mi_debug_set_spline_chunk(hook)
                got = ndi.spline_filter(xd, order=order, mode=mode).get()
                err = np.abs(got - ref).max() / np.abs(ref).max()
                if err > 1e-13: print("MISMATCH", shape, order, mode, hook, err)
    lib.mi_debug_set_spline_chunk(0)
    x32 = x.astype(np.float32); xd32 = ca.asarray(x32)
    ref = sndi.rotate(x32.astype(np.float64), 13.0, axes=(-1, -2), order=3, reshape=False, mode="mirror")
    got = ndi.rotate(xd32, 13.0, axes=(-1, -2), order=3, reshape=False, mode="mirror").get()
    print(shape, "rotate f32 err", np.abs(got - ref).max() / np.abs(ref).max())
    print("ok", shape)
for shape in [(2048, 2048), (4096, 4096), (8192, 8192)]:
    x = rng.standard_normal(shape).astype(np.float32); xd = ca.asarray(x)
    xd64 = ca.asarray(x.astype(np.float64))
    for hook in (-1, 0):
        lib.mi_debug_set_spline_chunk(hook)
        t1 = timeit(lambda: ndi.spline_filter(xd64, order=3))
        t2 = timeit(lambda: ndi.rotate(xd, 13.0, order=3, reshape=False))
        t3 = timeit(lambda: ndi.affine_transform(xd, np.array([[0.98, 0.05], [-0.05, 0.98]]), offset=(3.0, -2.0), order=3))
        t4 = timeit(lambda: ndi.zoom(xd, 1.5, order=3))
        print(shape, "hook", hook, "spline_filter f64 %.0f us  rotate f32 %.0f us  affine f32 %.0f us zoom1.5 %.0f us" % (t1, t2, t3, t4))
