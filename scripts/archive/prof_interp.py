"""Runs config D' (affine_transform order 1, 512^3; default) or D (MAP=1: map_coordinates) a few times for rocprofv3.
env: VAR = mi_debug_set_interp_c1 value (0 round-2 kernels, 1 r3 kernels, 2 r3 without the wide stores), REPS."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
_lib.load().mi_debug_set_interp_c1(int(os.environ.get("VAR", "1")))
n = 512
x = fs.volume_f32((n, n, n)); xd = ca.asarray(x); out = ca.empty(xd.shape, np.float32)
M, off = fs.affine_case(n)
reps = int(os.environ.get("REPS", "5"))
if os.environ.get("MAP"):
    cd = ca.asarray(fs.affine_coords_f32(n))
    for _ in range(reps):
        ndi.map_coordinates(xd, cd, order=1, mode="constant", output=out)
else:
    for _ in range(reps):
        ndi.affine_transform(xd, M, off, order=1, mode="constant", output=out)
ca.synchronize()
