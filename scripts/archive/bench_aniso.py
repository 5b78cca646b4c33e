"""3-D volumes with per-axis sizes / sigmas (anisotropic voxels, slice-wise filtering): time and algorithmic GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi
from cupyimg_amd import _lib
if 'IMG2D' in os.environ:
    _lib.load().mi_debug_set_sep3d_image2d(int(os.environ['IMG2D']))

def timeit(fn, reps=10):
    for _ in range(3): fn()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize()
    return e0.elapsed_ms(e1) / reps * 1e3

shape = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (512, 512, 512)
x = ca.asarray(np.random.default_rng(0).standard_normal(shape, dtype=np.float32))
u = ca.asarray(np.random.default_rng(1).integers(0, 256, size=shape, dtype=np.uint8))
o = ca.empty(shape, np.float32); uo = ca.empty(shape, np.uint8)
n = float(np.prod(shape))
print("shape", shape)
for size in [(5, 5, 5), (1, 5, 5), (3, 5, 5), (5, 5, 1), (5, 1, 5), (1, 1, 5), (5, 1, 1), (1, 9, 9), (3, 7, 7), (9, 9, 9), (1, 3, 3)]:
    t = timeit(lambda: ndi.uniform_filter(x, size=size, output=o))
    print("  uniform %-12s %8.1f us %6.0f GB/s %5.1f %%" % (size, t, 8 * n / t / 1e3, 8 * n / t / 1e3 / 80), flush=True)
for sigma in [(2, 2, 2), (0, 2, 2), (1, 2, 2), (2, 1, 1), (1, 1, 1), (0, 1, 1), (0.5, 1, 1), (0, 4, 4)]:
    t = timeit(lambda: ndi.gaussian_filter(x, sigma, output=o))
    print("  gauss   %-12s %8.1f us %6.0f GB/s %5.1f %%" % (sigma, t, 8 * n / t / 1e3, 8 * n / t / 1e3 / 80), flush=True)
for size in [(5, 5, 5), (1, 5, 5), (3, 5, 5), (1, 3, 3), (3, 3, 1)]:
    t = timeit(lambda: ndi.grey_erosion(x, size=size, output=o))
    print("  erode f32 %-10s %8.1f us %6.0f GB/s %5.1f %%" % (size, t, 8 * n / t / 1e3, 8 * n / t / 1e3 / 80), flush=True)
    t = timeit(lambda: ndi.grey_erosion(u, size=size, output=uo))
    print("  erode u8  %-10s %8.1f us %6.0f GB/s %5.1f %%" % (size, t, 2 * n / t / 1e3, 2 * n / t / 1e3 / 80), flush=True)
for size in [(1, 3, 3), (3, 3, 3)]:
    t = timeit(lambda: ndi.median_filter(x, size=size, output=o), reps=3)
    print("  median f32 %-9s %8.1f us %6.0f GB/s %5.1f %%" % (size, t, 8 * n / t / 1e3, 8 * n / t / 1e3 / 80), flush=True)
