"""Where do the device's order-3 results lose bit-equality with SciPy?  Stage by stage (run on the GPU box):
spline_filter1d per axis, spline_filter, shift on SciPy's own coefficients (prefilter=False), full shift."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.ndimage as sndi
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi
rng = np.random.default_rng(5)
for shape, sh, mode in [((2, 36, 264), [1.5, 0.0, 0.0], 'grid-wrap'), ((2, 36, 264), [1.5, 0.0, 0.0], 'reflect'), ((12, 2, 260), [0.0, 1.5, 0.0], 'wrap'), ((21, 2), [0.0, -2.25], 'mirror'), ((2, 1, 29), [1.5, 0.3, 0.0], 'mirror'),
                        ((8, 2, 520), [0.0, 1.5, 0.0], 'mirror'), ((9, 7, 33), [0.5, 1.5, 0.25], 'reflect')]:
    x = rng.integers(-200, 250, size=shape).astype(np.float64)
    xd = ca.asarray(x)
    for ax in range(x.ndim):
        ref = sndi.spline_filter1d(x, 3, axis=ax, mode=mode)
        got = ndi.spline_filter1d(xd, 3, axis=ax, mode=mode).get()
        print(shape, mode, "spline_filter1d axis", ax, "n =", shape[ax], "bit mismatches", int((got != ref).sum()), "of", ref.size,
              "max ulp-ish", float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300)))
    ref = sndi.spline_filter(x, 3, mode=mode)
    got = ndi.spline_filter(xd, 3, mode=mode).get()
    print(shape, mode, "spline_filter      bit mismatches", int((got != ref).sum()), "of", ref.size)
    # interpolation stage alone: SciPy's coefficients on both sides
    refs = sndi.shift(ref, sh, order=3, mode=mode, prefilter=False)
    gots = ndi.shift(ca.asarray(ref), sh, order=3, mode=mode, prefilter=False).get()
    print(shape, mode, "shift on given coefficients: bit mismatches", int((gots != refs).sum()), "of", refs.size)
    reff = sndi.shift(x, sh, order=3, mode=mode)
    gotf = ndi.shift(xd, sh, order=3, mode=mode).get()
    print(shape, mode, "full shift (float64): bit mismatches", int((gotf != reff).sum()), "of", reff.size)
    xi = x.astype(np.int32)
    xu = np.abs(x).astype(np.uint16)
    print(shape, mode, 'full shift (uint16):  mismatches', int((ndi.shift(ca.asarray(xu), sh, order=3, mode=mode).get() != sndi.shift(xu, sh, order=3, mode=mode)).sum()))
    print(shape, mode, "full shift (int32):   mismatches", int((ndi.shift(ca.asarray(xi), sh, order=3, mode=mode).get() != sndi.shift(xi, sh, order=3, mode=mode)).sum()))
