import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi
rng = np.random.default_rng(0)
n = 512
th = np.deg2rad(7.0)
M = np.diag([1.02, 1, 1]) @ np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]])
c = (n - 1) / 2
off = np.array([c, c, c]) - M @ np.array([c, c, c]) + np.array([0.5, -1.25, 2.0])
x = ca.asarray(rng.standard_normal((n, n, n), dtype=np.float32))
for _ in range(4):
    y = ndi.affine_transform(x, M, offset=off, order=3)
ca.synchronize()
