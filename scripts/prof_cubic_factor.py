"""rocprofv3 subject: the order-3 interpolation kernels of round 5 alone (prefilter=False) and the default calls, 512^3 float32.
usage (through scripts/kstat_any.sh / pmc_any.sh): scripts/kstat_any.sh <tag> scripts/prof_cubic_factor.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
n = 512
x = np.random.default_rng(0).standard_normal((n,) * 3).astype(np.float32)
xd = ca.asarray(x); out = ca.empty(x.shape, np.float32)
M, off = fs.affine_case(n)
reps = int(os.environ.get("REPS", "12"))
for _ in range(reps):
    ndi.affine_transform(xd, M, off, order=3, prefilter=False, output=out)
a = np.deg2rad(30.0); M2 = np.array([[1.0, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
ctr = np.array([(n - 1) / 2] * 3)
for _ in range(reps):
    ndi.affine_transform(xd, M2, ctr - M2 @ ctr + np.array([0.3, 0, 0]), order=3, prefilter=False, output=out)
for _ in range(reps):
    ndi.rotate(xd, 7.0, reshape=False, output=out)
for _ in range(reps):
    ndi.rotate(xd, 7.0, axes=(1, 2), reshape=False, output=out)
for _ in range(reps):
    ndi.affine_transform(xd, M, off, order=3, output=out)
out.get()
