"""Runs the headline op a few times (for rocprofv3).  env: CFG, ZCH, N, SIZE, REPS"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi

n = int(os.environ.get("N", "512"))
size = int(os.environ.get("SIZE", "5"))
lib = _lib.load()
lib.mi_debug_set_sep3d_cfg(int(os.environ.get("CFG", "0")))
lib.mi_debug_set_sep3d_zchunks(int(os.environ.get("ZCH", "0")))
lib.mi_debug_set_sep3d_long(int(os.environ.get("LONG", "0")))
lib.mi_debug_set_sep3d_zrev(int(os.environ.get("ZREV", "1")))
lib.mi_debug_set_long_zchunks(int(os.environ.get("LZCH", "0")))
x = np.random.default_rng(0).standard_normal((n, n, n), dtype=np.float32)
xd = ca.asarray(x)
out = ca.empty(xd.shape, np.float32)
for _ in range(int(os.environ.get("REPS", "5"))):
    ndi.uniform_filter(xd, size=size, output=out)
ca.synchronize()
