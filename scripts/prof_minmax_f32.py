"""Runs minimum_filter(size) on a 512^3 float32 volume a few times (for rocprofv3).  env SIZE, REPS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi
n = int(os.environ.get("N", "512"))
size = int(os.environ.get("SIZE", "7"))
x = np.random.default_rng(0).standard_normal((n, n, n), dtype=np.float32)
xd = ca.asarray(x); out = ca.empty(xd.shape, np.float32)
for _ in range(int(os.environ.get("REPS", "10"))):
    ndi.minimum_filter(xd, size=size, output=out)
ca.synchronize()
