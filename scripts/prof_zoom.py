import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
rng = np.random.default_rng(0)
n = 512
x = ca.asarray(rng.standard_normal((n, n, n), dtype=np.float32))
hook = _lib.load().mi_debug_set_cubic_diag
hook.argtypes = [ctypes.c_int]
for on in (1, 0):
    hook(on)
    for _ in range(3):
        y = ndi.zoom(x, 1.25, order=3)
        y = ndi.shift(x, 1.7, order=3)
    ca.synchronize()
