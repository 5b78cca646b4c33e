"""r5: the calls scripts/pmc_script.sh counts for the sorting-network rank kernel: 512^3 uint8 and float32, 27 samples, pruned
median and full network.   usage: bash scripts/pmc_script.sh r5_rank27 scripts/prof_rank27.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi
for dt in (np.uint8, np.float32):
    x = (np.random.default_rng(0).standard_normal((512, 512, 512)) * 50).astype(dt)
    xd = ca.asarray(x); out = ca.empty(x.shape, dt)
    for _ in range(2):
        ndi.median_filter(xd, size=3, output=out)
        ndi.rank_filter(xd, 8, size=3, output=out)
    ca.synchronize()
