"""Rank / median / percentile filters."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.ndimage as sndi
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi

def timeit(fn, reps=3):
    for _ in range(1): fn()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize()
    return e0.elapsed_ms(e1) / reps

rng = np.random.default_rng(0)
x3 = rng.standard_normal((256, 256, 256), dtype=np.float32)
u3 = rng.integers(0, 256, size=(256, 256, 256), dtype=np.uint8)
x2 = rng.standard_normal((4096, 4096), dtype=np.float32)
u2 = rng.integers(0, 256, size=(4096, 4096), dtype=np.uint8)
for name, arr, size in [("f32 256^3", x3, 3), ("u8  256^3", u3, 3), ("f32 4096^2", x2, 3), ("f32 4096^2", x2, 5), ("u8  4096^2", u2, 3),
                        ("u8  4096^2", u2, 5), ("f32 4096^2", x2, 7)]:
    d = ca.asarray(arr)
    o = ca.empty(arr.shape, arr.dtype)
    t = timeit(lambda: ndi.median_filter(d, size=size, output=o))
    print("median_filter size %d %s: %9.3f ms  (%8.0f Mvox/s)" % (size, name, t, arr.size / t / 1e3), flush=True)
sub = x3[:64]
t0 = time.perf_counter(); sndi.median_filter(sub, size=3); tc = time.perf_counter() - t0
print("scipy median 3^3 on 64x256x256 f32: %.2f s (%.1f Mvox/s)" % (tc, sub.size / tc / 1e6))
