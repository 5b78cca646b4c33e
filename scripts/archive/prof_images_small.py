"""Three launches of a few image kernels on 8192^2 (for the PMC passes of scripts/pmc_any.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi

n = 8192
rng = np.random.default_rng(0)
x = ca.asarray(rng.standard_normal((n, n), dtype=np.float32)); o = ca.empty((n, n), np.float32)
u = ca.asarray(rng.integers(0, 256, size=(n, n), dtype=np.uint8)); uo = ca.empty((n, n), np.uint8)
d = ca.asarray(rng.standard_normal((n, n))); do = ca.empty((n, n), np.float64)
for _ in range(3):
    ndi.uniform_filter(x, 5, output=o); ndi.gaussian_filter(x, 2.0, output=o)
    ndi.median_filter(x, 3, output=o); ndi.median_filter(u, 3, output=uo)
    ndi.grey_erosion(u, size=7, output=uo); ndi.gaussian_filter(d, 2.0, output=do)
ca.synchronize()
