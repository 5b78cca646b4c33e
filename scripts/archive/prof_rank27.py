"""r5: the calls scripts/pmc_script.sh / kstat_any.sh count for the 27-sample rank kernels on 512^3 (uint8 and float32): the
3 x 3 x 3 median on the kernel that shares its sorting between windows (median27_stream_kernel), the same on the per-voxel network
(mi_debug_set_median27(0): rank3_sorted_kernel<32,27,13>), and rank 8 of 27 (full 32-wire network).
usage: bash scripts/pmc_script.sh r5_rank27 scripts/prof_rank27.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
lib = _lib.load()
for dt in (np.uint8, np.float32):
    x = (np.random.default_rng(0).standard_normal((512, 512, 512)) * 50).astype(dt)
    xd = ca.asarray(x); out = ca.empty(x.shape, dt)
    for _ in range(3):
        ndi.median_filter(xd, size=3, output=out)
        lib.mi_debug_set_median27(0)
        ndi.median_filter(xd, size=3, output=out)
        lib.mi_debug_set_median27(1)
        ndi.rank_filter(xd, 8, size=3, output=out)
    ca.synchronize()
