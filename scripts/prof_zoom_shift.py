"""rocprofv3 subject: default-order zoom / shift of a 512^3 (zoom: 400^3 -> 512^3) float32 volume.  usage: scripts/kstat_any.sh <tag> scripts/prof_zoom_shift.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi
x = np.random.default_rng(0).standard_normal((512,) * 3).astype(np.float32)
xd = ca.asarray(x); out = ca.empty(x.shape, np.float32)
for _ in range(10):
    ndi.shift(xd, (0.5, -0.25, 0.75), output=out)
y = ca.asarray(x[:400, :400, :400].copy())
for _ in range(10):
    ndi.zoom(y, 1.28, output=out)
out.get()
