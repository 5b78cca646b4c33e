"""r3: config D' -- LDS-staged affine kernel (knob 1) against the L1-gather kernel (knob 5), interleaved long runs
(the clocks of the box move by 10 % with what ran before; only same-process ratios are comparable)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
lib = _lib.load()
n = 512
x = fs.volume_f32((n, n, n)); xd = ca.asarray(x); out = ca.empty(xd.shape, np.float32)
M, off = fs.affine_case(n)
def t(fn, reps=150):
    for _ in range(30): fn()
    ca.synchronize(); e0, e1 = ca.Event(), ca.Event(); e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize(); return e0.elapsed_ms(e1) / reps * 1e3
f = lambda: ndi.affine_transform(xd, M, off, order=1, mode="constant", output=out)
res = {1: [], 5: []}
for rep in range(4):
    for var in (5, 1):
        lib.mi_debug_set_interp_c1(var)
        res[var].append(t(f))
lib.mi_debug_set_interp_c1(1)
for var in (5, 1):
    print("knob %d (%s): %s us, median %.1f" % (var, "L1 gathers" if var == 5 else "LDS-staged", " ".join("%.1f" % v for v in res[var]), float(np.median(res[var]))), flush=True)
print("ratio LDS / L1: %.3f" % (np.median(res[1]) / np.median(res[5])))
