"""r6: grey erosion / min / max with a cubic size on float32 volumes whose rows are not a multiple of 16 bytes: the ragged
kernel (rows as they are) against the extended-rows route (mi_debug_set_sep3d_ragged(0)) and uniform_filter of the same size.
-> profiles/r6_ragged_minmax.txt"""
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
for shape in ((181, 217, 181), (91, 109, 91), (193, 229, 193), (256, 256, 255), (300, 300, 301)):
    x = ca.asarray(rng.standard_normal(shape).astype(np.float32)); out = ca.empty(shape, np.float32)
    for size in (3, 5, 7, 9):
        t1, _ = timeit(lambda: ndi.grey_erosion(x, size=size, output=out), 10); k = last_kernel()[4:60]
        lib.mi_debug_set_sep3d_ragged(0)
        t0, _ = timeit(lambda: ndi.grey_erosion(x, size=size, output=out), 10)
        lib.mi_debug_set_sep3d_ragged(1)
        tu, _ = timeit(lambda: ndi.uniform_filter(x, size, output=out), 10)
        print("%-16s grey_erosion %d: extended rows %7.1f us -> ragged kernel %7.1f us (%.3f of 8 TB/s)   uniform_filter %d %7.1f us   %s" % (
            shape, size, t0 * 1e6, t1 * 1e6, 2 * x.nbytes / 8e12 / t1, size, tu * 1e6, k), flush=True)
    del x, out; ca.free_all_blocks()
# uint8: mm3u8_ragged_kernel (rows as they lie) against the extended-rows route (mi_debug_set_u8_ragged(0))
lib.mi_debug_set_u8_ragged.argtypes = [ctypes.c_int]
for shape in ((181, 217, 181), (91, 109, 91), (193, 229, 193), (256, 256, 255), (300, 300, 301), (400, 400, 401)):
    x = ca.asarray(rng.integers(0, 256, size=shape).astype(np.uint8)); out = ca.empty(shape, np.uint8)
    for size in (3, 5, 7):
        t1, _ = timeit(lambda: ndi.grey_erosion(x, size=size, output=out), 10); k = last_kernel()[4:50]
        lib.mi_debug_set_u8_ragged(0)
        t0, _ = timeit(lambda: ndi.grey_erosion(x, size=size, output=out), 10)
        lib.mi_debug_set_u8_ragged(1)
        print("%-16s uint8 grey_erosion %d: extended rows %7.1f us -> as they lie %7.1f us (%.3f of 8 TB/s)   %s" % (
            shape, size, t0 * 1e6, t1 * 1e6, 2 * x.nbytes / 8e12 / t1, k), flush=True)
    del x, out; ca.free_all_blocks()
