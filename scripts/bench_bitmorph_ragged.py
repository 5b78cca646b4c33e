"""r6: binary morphology on bool volumes whose rows are NOT a multiple of 16 bytes (MNI grids): the bit kernel on the rows as they are\nagainst what ran before.  -> profiles/r6_binary_ragged.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import ctypes
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
from bench_bitmorph import timeit
lib = _lib.load(); lib.mi_debug_set_bitmorph_ragged.argtypes = [ctypes.c_int]
rng = np.random.default_rng(0)
print("# binary morphology on bool volumes whose rows are not a multiple of 16 bytes: before (extended rows for one iteration, generic kernel otherwise: mi_debug_set_bitmorph_ragged(0)) -> bit kernel on the rows as they are")
for shape in ((181, 217, 181), (91, 109, 91), (193, 229, 193), (256, 256, 255), (192, 224, 192)):
    b = ca.asarray(rng.random(shape) > 0.3); bo = ca.empty(shape, bool)
    for name, fn in [("erosion", lambda: ndi.binary_erosion(b, output=bo)), ("erosion x3", lambda: ndi.binary_erosion(b, iterations=3, output=bo)),
                     ("dilation 5^3", lambda: ndi.binary_dilation(b, np.ones((5,5,5)), output=bo)), ("opening", lambda: ndi.binary_opening(b, output=bo)),
                     ("fill_holes", lambda: ndi.binary_fill_holes(b))]:
        lib.mi_debug_set_bitmorph_ragged(0); t0 = timeit(fn, 5.0)
        lib.mi_debug_set_bitmorph_ragged(1); t1 = timeit(fn, 5.0)
        print("%-16s %-13s %9.1f us -> %8.1f us   %s" % (shape, name, t0, t1, ca.last_kernel()[4:52]), flush=True)
