"""End-to-end consumer of the filter path: structural similarity of two
float32 volumes, device-resident, against SciPy on the host for a bounded sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.ndimage as sndi
import cupyimg_amd as ca
from cupyimg_amd.skimage import metrics

def timeit(fn, reps=5):
    for _ in range(2): fn()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize()
    return e0.elapsed_ms(e1) / reps

def host_ssim(x, y, win=7):
    f = lambda a: sndi.uniform_filter(a, size=win, mode="reflect")
    NP = win ** x.ndim; cov = NP / (NP - 1)
    ux, uy = f(x), f(y); uxx, uyy, uxy = f(x * x), f(y * y), f(x * y)
    vx, vy, vxy = cov * (uxx - ux * ux), cov * (uyy - uy * uy), cov * (uxy - ux * uy)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    p = win // 2
    return S[p:-p, p:-p, p:-p].mean(dtype=np.float64)

rng = np.random.default_rng(0)
for n in (256, 512):
    x = rng.random((n, n, n), dtype=np.float32)
    y = (x + 0.05 * rng.standard_normal(x.shape, dtype=np.float32)).astype(np.float32)
    xd, yd = ca.asarray(x), ca.asarray(y)
    for kw, name in [({}, "uniform 7"), ({"gaussian_weights": True}, "gaussian 11")]:
        t = timeit(lambda: metrics.structural_similarity(xd, yd, data_range=1.0, data_dtype=np.float32, **kw), 5)
        print("ssim %-11s f32 %d^3: %7.3f ms  (%8.0f Mvox/s; 12 B/vox algorithmic -> %5.1f%% of 8 TB/s)" % (
            name, n, t, n ** 3 / t / 1e3, 12 * n ** 3 / t / 1e6 / 80), flush=True)
    if n == 256:
        got = metrics.structural_similarity(xd, yd, data_range=1.0, data_dtype=np.float32)
        t0 = time.perf_counter(); want = host_ssim(x, y); tc = time.perf_counter() - t0
        print("   host SciPy/NumPy 256^3: %.2f s (%.1f Mvox/s); mssim device %.7f host %.7f" % (tc, n ** 3 / tc / 1e6, got, want))
    xd = yd = None
    ca.free_all_blocks()
