"""Host-side cost of one API call (small image, so the kernel is negligible): wall time per call and a cProfile listing."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi

x = ca.asarray(np.random.default_rng(0).standard_normal((256, 256), dtype=np.float32))
u = ca.asarray(np.random.default_rng(1).integers(0, 256, size=(256, 256), dtype=np.uint8))
v = ca.asarray(np.random.default_rng(0).standard_normal((32, 32, 32), dtype=np.float32))
o = ca.empty((256, 256), np.float32); uo = ca.empty((256, 256), np.uint8); vo = ca.empty((32, 32, 32), np.float32)
k33 = np.ones((3, 3), np.float32)
cases = [("uniform5 2d", lambda: ndi.uniform_filter(x, size=5, output=o)),
         ("uniform5 3d", lambda: ndi.uniform_filter(v, size=5, output=vo)),
         ("uniform5 3d alloc", lambda: ndi.uniform_filter(v, size=5)),
         ("gauss2 2d", lambda: ndi.gaussian_filter(x, 2.0, output=o)),
         ("erode3 u8", lambda: ndi.grey_erosion(u, size=3, output=uo)),
         ("erode3 f32 3d", lambda: ndi.grey_erosion(v, size=3, output=vo)),
         ("median3", lambda: ndi.median_filter(x, size=3, output=o)),
         ("sobel", lambda: ndi.sobel(x, output=o)),
         ("corr3x3", lambda: ndi.correlate(x, k33, output=o)),
         ("affine", lambda: ndi.affine_transform(v, np.eye(3), order=1, output=vo)),
         ]
N = 2000
for name, fn in cases:
    for _ in range(20): fn()
    ca.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): fn()
    t1 = time.perf_counter()
    ca.synchronize()
    t2 = time.perf_counter()
    print("%-18s issue %6.1f us/call   with sync %6.1f us/call" % (name, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6), flush=True)
which = os.environ.get("PROFILE", "uniform5 2d")
fn = dict(cases)[which]
pr = cProfile.Profile()
pr.enable()
for _ in range(N): fn()
pr.disable()
ca.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
