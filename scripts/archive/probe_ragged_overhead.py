import json, os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit
lib = _lib.load()
rng = np.random.default_rng(0)
for shape in ((182, 218, 184), (181, 217, 181), (181, 217, 184), (182, 218, 181)):
    x = rng.standard_normal(shape).astype(np.float32)
    xd = ca.asarray(x); out = ca.empty(shape, np.float32)
    for size in (3, 5):
        for knob in (1, 2):
            lib.mi_debug_set_sep3d_ragged(knob)
            t, _ = timeit(lambda: ndi.uniform_filter(xd, size, output=out), 30)
            lib.mi_debug_set_sep3d_ragged(1)
            print(shape, size, knob, round(t * 1e6, 1), last_kernel()[4:60], flush=True)
