"""skimage facade on images: gaussian (float64 default), morphology with the default / disk footprints, warp / rotate-like maps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.skimage import filters as skf, morphology as skm, transform as skt

def timeit(fn, reps=5):
    for _ in range(2): fn()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize()
    return e0.elapsed_ms(e1) / reps * 1e3

shape = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (4096, 4096)
rng = np.random.default_rng(0)
u = ca.asarray(rng.integers(0, 256, size=shape, dtype=np.uint8))
f = ca.asarray(rng.random(shape, dtype=np.float32))
d = ca.asarray(rng.random(shape))
b = ca.asarray(rng.random(shape) > 0.4)
def affine_map(coords):
    return coords * 0.98 + 3.0
ops = [("gaussian(u8) s=2", lambda: skf.gaussian(u, 2.0)), ("gaussian(f32) s=2", lambda: skf.gaussian(f, 2.0)), ("gaussian(f64) s=2", lambda: skf.gaussian(d, 2.0)),
       ("erosion(u8) default", lambda: skm.erosion(u)), ("erosion(u8) disk3", lambda: skm.erosion(u, skm.disk(3))), ("dilation(u8) disk2", lambda: skm.dilation(u, skm.disk(2))),
       ("opening(u8) disk2", lambda: skm.opening(u, skm.disk(2))), ("white_tophat(u8) disk2", lambda: skm.white_tophat(u, skm.disk(2))),
       ("erosion(f32) disk2", lambda: skm.erosion(f, skm.disk(2))), ("binary_erosion default", lambda: skm.binary_erosion(b)),
       ("binary_opening disk2", lambda: skm.binary_opening(b, skm.disk(2)))]
for name, fn in ops:
    try:
        t = timeit(fn)
        print("   %-26s %9.1f us  %7.0f Mpix/s" % (name, t, np.prod(shape) / t), flush=True)
    except Exception as exc:
        print("   %-26s FAILED %s %s" % (name, type(exc).__name__, str(exc)[:100]), flush=True)
