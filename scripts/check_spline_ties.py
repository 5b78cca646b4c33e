import sys
sys.path.insert(0, "/root/repo")
import numpy as np, scipy.ndimage as sndi
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi
from cupyimg_amd import _lib
lib = _lib.load()
rng = np.random.default_rng(5)
cases = [((12, 2, 260), 'int32', [0.0, 1.5, 0.0], 'wrap'), ((2, 1, 29), 'uint16', [1.5, 0.3, 0.0], 'mirror'), ((21, 2), 'int32', [0.0, -2.25], 'mirror'), ((8, 2, 520), 'int32', [0.0, 1.5, 0.0], 'mirror')]
for hook in (0, 1):
    lib.mi_debug_set_spline_gain_first(hook)
    tot = 0
    for shape, dt, sh, mode in cases:
        for rep in range(20):
            x = rng.integers(-200 if dt == 'int32' else 0, 250, size=shape).astype(dt)
            ref = sndi.shift(x, sh, order=3, mode=mode, cval=2.0)
            got = ndi.shift(ca.asarray(x), sh, order=3, mode=mode, cval=2.0).get()
            tot += int((got != ref).sum())
    print("gain_first", hook, "mismatching outputs over the tie cases:", tot)
