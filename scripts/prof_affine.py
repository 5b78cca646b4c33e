"""Runs config D' (affine_transform order 1, 512^3) or, with MAP=1, config D (map_coordinates) a few times (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi
n = 512
x = np.random.default_rng(0).standard_normal((n, n, n), dtype=np.float32)
xd = ca.asarray(x); out = ca.empty(xd.shape, np.float32)
ang = np.deg2rad(7.0)
R = np.array([[1, 0, 0], [0, np.cos(ang), -np.sin(ang)], [0, np.sin(ang), np.cos(ang)]])
M = np.diag([1.02, 1.0, 1.0]) @ R
ctr = (n - 1) / 2.0
off = ctr - M @ np.array([ctr] * 3) + np.array([0.5, -1.25, 2.0])
if os.environ.get("MAP"):
    idx = np.indices((n, n, n), dtype=np.float32).reshape(3, -1)
    coords = (M.astype(np.float32) @ idx + off.astype(np.float32)[:, None]).reshape(3, n, n, n)
    cd = ca.asarray(coords)
    for _ in range(5):
        ndi.map_coordinates(xd, cd, order=1, mode="constant", output=out)
else:
    for _ in range(5):
        ndi.affine_transform(xd, M, off, order=1, mode="constant", output=out)
ca.synchronize()
