"""Runs config B (gaussian sigma=2, 512^3) and the headline (uniform 5) a few dozen times for rocprofv3 counter passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
n = 512
x = fs.volume_f32((n, n, n)); xd = ca.asarray(x); out = ca.empty(xd.shape, np.float32)
for _ in range(int(os.environ.get("REPS", "12"))):
    ndi.gaussian_filter(xd, 2.0, output=out)
for _ in range(int(os.environ.get("REPS", "12"))):
    ndi.uniform_filter(xd, 5, output=out)
ca.synchronize()
