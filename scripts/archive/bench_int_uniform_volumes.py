import sys
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
for shape in [(256, 256, 256), (512, 512, 512)]:
    for dt in ("uint8", "int16"):
        x = ca.asarray(np.random.default_rng(0).integers(0, 200, size=shape).astype(dt)); o = ca.empty(shape, np.dtype(dt))
        print(shape, dt, "uniform 5: %.1f us   uniform (1,5,5): %.1f us   uniform 9: %.1f us" % (timeit(lambda: ndi.uniform_filter(x, 5, output=o)), timeit(lambda: ndi.uniform_filter(x, (1, 5, 5), output=o)), timeit(lambda: ndi.uniform_filter(x, 9, output=o))), flush=True)
