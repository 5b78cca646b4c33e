"""r3: kernel generations of the 17-tap long kernel under the settled protocol (library built with MI_LONG_TUNE=1):
mi_debug_set_long_rows 0 = r3 stream, 1 = r2 stream, 2 = r2 stream with two output rows per wave."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
lib = _lib.load()
n = 512
x = fs.volume_f32((n, n, n)); xd = ca.asarray(x); o = ca.empty((n, n, n), np.float32)
def t(fn, reps=250):
    for _ in range(180): fn()
    ca.synchronize(); e0, e1 = ca.Event(), ca.Event(); e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize(); return e0.elapsed_ms(e1) / reps * 1e3
f = lambda: ndi.gaussian_filter(xd, 2.0, output=o)
outs = {}
for rows in (0, 1, 2):
    lib.mi_debug_set_long_rows(rows); f(); outs[rows] = o.get()
print("max |r2 - two rows| %.1e, max |r3 - r2| %.1e" % (np.abs(outs[1] - outs[2]).max(), np.abs(outs[0] - outs[1]).max()))
res = {0: [], 1: [], 2: []}
for rep in range(3):
    for rows in (0, 1, 2):
        lib.mi_debug_set_long_rows(rows)
        res[rows].append(t(f))
lib.mi_debug_set_long_rows(0)
for rows, name in ((0, "r3 stream"), (1, "r2 stream"), (2, "r2 stream, two rows per wave")):
    print("%-30s %s us" % (name, " ".join("%.1f" % v for v in res[rows])), flush=True)
