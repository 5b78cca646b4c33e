"""r6: what the common calls cost on an MNI-grid volume (181 x 217 x 181: rows of 181 samples, not a multiple of 16 bytes) against the
padded 182 x 218 x 184 volume -- which calls still pay a detour for ragged rows.  -> profiles/r6_mni_survey.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd import last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit
rng = np.random.default_rng(0)
ang = np.deg2rad(7.0)
M = np.array([[1, 0, 0], [0, np.cos(ang), -np.sin(ang)], [0, np.sin(ang), np.cos(ang)]])
def calls(x, b, u, i16):
    ctr = (np.array(x.shape) - 1) / 2.0
    off = ctr - M @ ctr
    return [
        ("gaussian_filter sigma 1.0", lambda: ndi.gaussian_filter(x, 1.0)),
        ("gaussian_filter sigma 2.0", lambda: ndi.gaussian_filter(x, 2.0)),
        ("gaussian_filter sigma (1, 1.5, 2)", lambda: ndi.gaussian_filter(x, (1.0, 1.5, 2.0))),
        ("uniform_filter 3", lambda: ndi.uniform_filter(x, 3)),
        ("uniform_filter 9", lambda: ndi.uniform_filter(x, 9)),
        ("median_filter 3", lambda: ndi.median_filter(x, 3)),
        ("correlate 3x3x3", lambda: ndi.correlate(x, W3)),
        ("sobel axis 0", lambda: ndi.sobel(x, 0)),
        ("gaussian_gradient_magnitude 1.5", lambda: ndi.gaussian_gradient_magnitude(x, 1.5)),
        ("laplace", lambda: ndi.laplace(x)),
        ("grey_erosion 3 (float32)", lambda: ndi.grey_erosion(x, size=3)),
        ("grey_erosion 9 (float32)", lambda: ndi.grey_erosion(x, size=9)),
        ("grey_erosion 3 (uint8)", lambda: ndi.grey_erosion(u, size=3)),
        ("grey_erosion 3 (int16)", lambda: ndi.grey_erosion(i16, size=3)),
        ("binary_erosion", lambda: ndi.binary_erosion(b)),
        ("binary_opening", lambda: ndi.binary_opening(b)),
        ("binary_dilation x3", lambda: ndi.binary_dilation(b, iterations=3)),
        ("binary_fill_holes", lambda: ndi.binary_fill_holes(b)),
        ("affine_transform order 1 (7 deg)", lambda: ndi.affine_transform(x, M, off, order=1)),
        ("affine_transform order 3 (7 deg)", lambda: ndi.affine_transform(x, M, off, order=3)),
        ("zoom 1.5 order 1", lambda: ndi.zoom(x, 1.5, order=1)),
        ("zoom 1.5 order 3", lambda: ndi.zoom(x, 1.5, order=3)),
        ("shift (0.5, 1.25, -2) order 3", lambda: ndi.shift(x, (0.5, 1.25, -2.0), order=3)),
        ("spline_filter order 3", lambda: ndi.spline_filter(x, 3)),
    ]
W3 = rng.standard_normal((3, 3, 3))
res = {}
for shape in ((181, 217, 181), (182, 218, 184)):
    g = np.indices(shape).astype(np.float32)
    r2 = sum(((g[i] - (shape[i] - 1) / 2) / (0.4 * shape[i])) ** 2 for i in range(3))
    x = ca.asarray(rng.standard_normal(shape).astype(np.float32))
    b = ca.asarray((r2 < 1.0) & (rng.random(shape) > 0.02))
    u = ca.asarray(rng.integers(0, 256, size=shape).astype(np.uint8))
    i16 = ca.asarray(rng.integers(-2000, 2000, size=shape).astype(np.int16))
    for name, f in calls(x, b, u, i16):
        try:
            t, _ = timeit(f, 10)
            res.setdefault(name, []).append((t * 1e6, last_kernel()[4:56]))
        except Exception as e:
            res.setdefault(name, []).append((float("nan"), repr(e)[:50]))
    del x, b, u, i16; ca.free_all_blocks()
print("%-36s %12s %12s   kernel of the last launch (181 x 217 x 181)" % ("call, whole call incl. allocation", "181x217x181", "182x218x184"))
for name, v in res.items():
    print("%-36s %9.1f us %9.1f us   %s" % (name, v[0][0], v[1][0], v[0][1]))
