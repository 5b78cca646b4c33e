"""3-D morphology on masks / uint8 volumes with the default (6-connected) structure, 3^3 and ball footprints."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi

def timeit(fn, reps=5):
    for _ in range(2): fn()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize()
    return e0.elapsed_ms(e1) / reps * 1e3

def ball(r):
    z, y, x = np.mgrid[-r:r + 1, -r:r + 1, -r:r + 1]
    return (x * x + y * y + z * z) <= r * r

for shape in [(256, 256, 256), (512, 512, 512)]:
    rng = np.random.default_rng(0)
    b = ca.asarray(rng.random(shape) > 0.3); bo = ca.empty(shape, bool)
    u = ca.asarray(rng.integers(0, 256, size=shape, dtype=np.uint8)); uo = ca.empty(shape, np.uint8)
    f = ca.asarray(rng.standard_normal(shape, dtype=np.float32)); fo = ca.empty(shape, np.float32)
    n = float(np.prod(shape))
    cross = ndi.generate_binary_structure(3, 1); full = np.ones((3, 3, 3), bool)
    print("shape", shape)
    for name, fn, isz in [("binary_erosion cross", lambda: ndi.binary_erosion(b, output=bo), 1), ("binary_erosion 3^3", lambda: ndi.binary_erosion(b, structure=full, output=bo), 1),
                          ("binary_dilation ball2", lambda: ndi.binary_dilation(b, structure=ball(2), output=bo), 1), ("binary_erosion cross x3", lambda: ndi.binary_erosion(b, iterations=3, output=bo), 1),
                          ("binary_opening cross", lambda: ndi.binary_opening(b, output=bo), 1),
                          ("grey_erosion u8 cross", lambda: ndi.grey_erosion(u, footprint=cross, output=uo), 1), ("grey_erosion u8 ball2", lambda: ndi.grey_erosion(u, footprint=ball(2), output=uo), 1),
                          ("grey_erosion u8 size3", lambda: ndi.grey_erosion(u, size=3, output=uo), 1),
                          ("grey_erosion f32 cross", lambda: ndi.grey_erosion(f, footprint=cross, output=fo), 4), ("grey_erosion f32 ball2", lambda: ndi.grey_erosion(f, footprint=ball(2), output=fo), 4),
                          ("median f32 cross", lambda: ndi.median_filter(f, footprint=cross, output=fo), 4)]:
        t = timeit(fn)
        print("   %-26s %9.1f us %6.0f GB/s  %4.1f %%" % (name, t, 2 * isz * n / t / 1e3, 2 * isz * n / t / 1e3 / 80.0), flush=True)
    b = bo = u = uo = f = fo = None
    ca.free_all_blocks()
