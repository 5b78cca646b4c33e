"""A few launches of each image kernel (for rocprofv3 --kernel-trace): 8192^2 images, 5 calls each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi

n = int(os.environ.get("N", "8192"))
rng = np.random.default_rng(0)
x = ca.asarray(rng.standard_normal((n, n), dtype=np.float32)); o = ca.empty((n, n), np.float32)
d = ca.asarray(rng.standard_normal((n, n))); do = ca.empty((n, n), np.float64)
u = ca.asarray(rng.integers(0, 256, size=(n, n), dtype=np.uint8)); uo = ca.empty((n, n), np.uint8)
h = ca.asarray(rng.integers(0, 65536, size=(n, n), dtype=np.uint16)); ho = ca.empty((n, n), np.uint16)
for _ in range(5):
    ndi.uniform_filter(x, 5, output=o); ndi.gaussian_filter(x, 1.0, output=o); ndi.gaussian_filter(x, 2.0, output=o)
    ndi.median_filter(x, 3, output=o); ndi.median_filter(u, 3, output=uo); ndi.median_filter(h, 3, output=ho)
    ndi.grey_erosion(u, size=7, output=uo); ndi.grey_erosion(u, size=3, output=uo); ndi.grey_erosion(h, size=5, output=ho)
    ndi.uniform_filter(d, 5, output=do); ndi.gaussian_filter(d, 2.0, output=do); ndi.grey_erosion(d, size=5, output=do)
    ndi.affine_transform(d, np.array([[0.98, 0.05], [-0.05, 0.98]]), offset=(3.0, -2.0), order=1, output=do)
    ndi.spline_filter(d, order=3, output=do)
ca.synchronize()
