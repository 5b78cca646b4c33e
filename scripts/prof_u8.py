"""Runs config C (grey_erosion size 7 on 1024^3 uint8) a few times (for rocprofv3).  env SIZE, REPS, U8CFG, N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
lib = _lib.load()
lib.mi_debug_set_u8_fused(int(os.environ.get("U8CFG", "1")))
n = int(os.environ.get("N", "1024"))
size = int(os.environ.get("SIZE", "7"))
u = np.random.default_rng(1).integers(0, 256, size=(n, n, n), dtype=np.uint8)
ud = ca.asarray(u); uo = ca.empty(ud.shape, np.uint8)
for _ in range(int(os.environ.get("REPS", "12"))):
    ndi.grey_erosion(ud, size=size, output=uo)
ca.synchronize()
