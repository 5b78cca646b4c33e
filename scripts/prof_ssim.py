"""A few SSIM calls on 512^3 float32 (for rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.skimage import metrics
rng = np.random.default_rng(0)
n = int(os.environ.get("N", "512"))
x = rng.random((n, n, n), dtype=np.float32)
y = (x + 0.05 * rng.standard_normal((n, n, n), dtype=np.float32)).astype(np.float32)
xd, yd = ca.asarray(x), ca.asarray(y)
for _ in range(4):
    metrics.structural_similarity(xd, yd, data_range=1.0, data_dtype=np.float32)
ca.synchronize()
