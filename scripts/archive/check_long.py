"""Parity of the fused long-kernel path (sep3d_long.hip) against scipy.ndimage on assorted shapes / modes / origins."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.ndimage as sndi
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi

rng = np.random.default_rng(3)
bad = 0
cases = 0
shapes = [(40, 48, 256), (33, 21, 260), (70, 100, 512), (20, 16, 64), (17, 50, 1024), (9, 9, 16), (5, 70, 300), (130, 40, 252), (256, 256, 256)]
for shape in shapes:
    x = rng.standard_normal(shape).astype(np.float32)
    xd = ca.asarray(x)
    for mode in ["reflect", "mirror", "nearest", "wrap"]:
        for size in (11, 13, 15, 17):
            want = sndi.uniform_filter(x.astype(np.float64), size, mode=mode)
            got = ndi.uniform_filter(xd, size, mode=mode).get()
            err = np.abs(got - want).max() / np.abs(want).max()
            cases += 1
            if err > 1e-6:
                bad += 1
                print("uniform", shape, mode, size, err, flush=True)
        for sigma in (1.3, 2.0):
            want = sndi.gaussian_filter(x.astype(np.float64), sigma, mode=mode)
            got = ndi.gaussian_filter(xd, sigma, mode=mode).get()
            err = np.abs(got - want).max() / np.abs(want).max()
            cases += 1
            if err > 1e-6:
                bad += 1
                print("gauss", shape, mode, sigma, err, flush=True)
    for mode, cval in [("constant", 0.0), ("constant", 2.5), (["constant", "reflect", "nearest"], -1.0), (["wrap", "constant", "constant"], 0.75)]:
        for size in (9, 11, 17):
            want = sndi.uniform_filter(x.astype(np.float64), size, mode=mode, cval=cval)
            got = ndi.uniform_filter(xd, size, mode=mode, cval=cval).get()
            err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
            cases += 1
            if err > 1e-6:
                bad += 1
                print("const uniform", shape, mode, cval, size, err, flush=True)
        want = sndi.gaussian_filter(x.astype(np.float64), 2.0, mode=mode, cval=cval)
        got = ndi.gaussian_filter(xd, 2.0, mode=mode, cval=cval).get()
        err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
        cases += 1
        if err > 1e-6:
            bad += 1
            print("const gauss", shape, mode, cval, err, flush=True)
    for origin in [(2, -3, 0), (-5, 5, 0)]:
        want = sndi.uniform_filter(x.astype(np.float64), 11, mode="reflect", origin=origin)
        got = ndi.uniform_filter(xd, 11, mode="reflect", origin=origin).get()
        err = np.abs(got - want).max() / np.abs(want).max()
        cases += 1
        if err > 1e-6:
            bad += 1
            print("origin", shape, origin, err, flush=True)
print("cases", cases, "bad", bad)
