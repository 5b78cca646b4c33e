import os, sys
sys.path.insert(0, "/root/repo")
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
def disk(r):
    y, x = np.mgrid[-r:r + 1, -r:r + 1]
    return (x * x + y * y) <= r * r
for shape in [(4096, 4096), (8192, 8192)]:
    b = ca.asarray(np.random.default_rng(0).random(shape) > 0.3); bo = ca.empty(shape, bool)
    u = b.astype(np.uint8); uo = ca.empty(shape, np.uint8)
    for r in (1, 2, 3, 4):
        t1 = timeit(lambda: ndi.binary_erosion(b, structure=disk(r), output=bo))
        t2 = timeit(lambda: ndi.binary_dilation(b, structure=disk(r), output=bo))
        t3 = timeit(lambda: ndi.grey_erosion(u, footprint=disk(r), output=uo))
        t4 = timeit(lambda: ndi.binary_erosion(b, structure=np.ones((2 * r + 1,) * 2, bool), output=bo))
        print(shape, "disk(%d): binary_erosion %.1f us  binary_dilation %.1f us   grey_erosion(u8 runs) %.1f us   binary_erosion square %.1f us" % (r, t1, t2, t3, t4), flush=True)
