"""r6: gaussian_filter with 11 / 13 / 17 taps on float32 volumes whose rows are not a multiple of 16 bytes: sep3d_long3_kernel<..., ragged>
(rows as they lie) against the extended-rows route (mi_debug_set_sep3d_ragged(0)); last shape: an aligned volume of the same size.
-> profiles/r6_ragged_long.txt"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit
lib = _lib.load()
rng = np.random.default_rng(0)
for shape in ((181, 217, 181), (91, 109, 91), (193, 229, 193), (256, 256, 255), (300, 300, 301), (182, 218, 184)):
    x = ca.asarray(rng.standard_normal(shape).astype(np.float32)); out = ca.empty(shape, np.float32)
    for sigma in (1.3, 1.5, 2.0):
        t1, _ = timeit(lambda: ndi.gaussian_filter(x, sigma, output=out), 10); k = last_kernel()[4:48]
        lib.mi_debug_set_sep3d_ragged(0)
        t0, _ = timeit(lambda: ndi.gaussian_filter(x, sigma, output=out), 10)
        lib.mi_debug_set_sep3d_ragged(1)
        print("%-16s gaussian_filter sigma %.1f: extended rows %7.1f us -> as they lie %7.1f us (%.3f of 8 TB/s)   %s" % (
            shape, sigma, t0 * 1e6, t1 * 1e6, 2 * x.nbytes / 8e12 / t1, k), flush=True)
    del x, out; ca.free_all_blocks()
