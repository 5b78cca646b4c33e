"""Binary erosion / dilation / opening on bool IMAGES: bit kernel (one-plane volumes) against the byte kernel.
-> profiles/r6_binary_images.txt"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_bitmorph import timeit
lib = _lib.load()
lib.mi_debug_set_bitmorph_2d.argtypes = [ctypes.c_int]
rng = np.random.default_rng(0)
disk = (np.indices((5, 5)) - 2); disk = (disk ** 2).sum(0) <= 4
for shape in ((512, 512), (1024, 1024), (2048, 2048), (4096, 4096), (8192, 8192), (3000, 4000)):
    b = ca.asarray(rng.random(shape) > 0.3); bo = ca.empty(shape, bool)
    n = float(np.prod(shape))
    for name, fn in [("erosion cross", lambda: ndi.binary_erosion(b, output=bo)), ("erosion 3x3", lambda: ndi.binary_erosion(b, np.ones((3, 3)), output=bo)),
                     ("dilation disk2", lambda: ndi.binary_dilation(b, disk, output=bo)), ("erosion cross x4", lambda: ndi.binary_erosion(b, iterations=4, output=bo)),
                     ("opening cross", lambda: ndi.binary_opening(b, output=bo))]:
        lib.mi_debug_set_bitmorph_2d(0); t0 = timeit(fn, 10.0)
        lib.mi_debug_set_bitmorph_2d(1); t1 = timeit(fn, 10.0)
        print("%-14s %-18s byte kernel %8.1f us   bit kernel %8.1f us (%.3f of 8 TB/s)  %s" % (shape, name, t0, t1, 2 * n / t1 / 1e6 / 8, last_kernel()[20:70]), flush=True)
    del b, bo; ca.free_all_blocks()
