"""r6: binary_fill_holes on smooth and on tortuous masks: the block-wise fill kernel (bitfill3_kernel: blocks swept in place, whole mask
runs filled along x per sweep) against global iteration in fused batches (mi_debug_set_bitfill(0)).  -> profiles/r6_fill_holes.txt"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import numpy as np
import scipy.ndimage as sndi
import cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
from bench_bitmorph import timeit
lib = _lib.load(); lib.mi_debug_set_bitfill.argtypes = [ctypes.c_int]
rng = np.random.default_rng(0)
print("# binary_fill_holes: global iteration in fused batches (mi_debug_set_bitfill(0)) -> block-wise fill (bitfill3_kernel)")
for shape in ((181, 217, 181), (256, 256, 256), (192, 224, 192), (512, 512, 512)):
    g = np.indices(shape).astype(np.float32)
    c = [(n - 1) / 2 for n in shape]
    r2 = sum(((g[i] - c[i]) / (0.42 * shape[i])) ** 2 for i in range(3))
    del g
    cases = {"ellipsoid with 2 % pinholes": (r2 < 1.0) & (rng.random(shape) > 0.02),
             "shell (one big cavity)": (r2 < 1.0) & (r2 > 0.5),
             "noise 70 % (tortuous)": rng.random(shape) > 0.3}
    for name, x in cases.items():
        xd = ca.asarray(x)
        ref = sndi.binary_fill_holes(x) if x.size <= 2 ** 25 else None
        res = []
        for on in (0, 1):
            lib.mi_debug_set_bitfill(on)
            got = ndi.binary_fill_holes(xd).get()
            ok = ref is None or np.array_equal(got, ref)
            res.append((timeit(lambda: ndi.binary_fill_holes(xd), 3.0), ok))
        print("%-16s fill_holes %-30s %9.1f us -> %9.1f us   parity %s %s   %s" % (shape, name, res[0][0], res[1][0], res[0][1], res[1][1], ca.last_kernel()[4:40]), flush=True)
        del xd
    ca.free_all_blocks()
