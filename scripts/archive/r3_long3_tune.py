"""r3: tuning variants of sep3d_long3_kernel<17> (library built with MI_LONG_TUNE=1): read-group sizes and the place
of the halo-table pass.  Interleaved repeats, sustained timings on config B (512^3, sigma=2)."""
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
    # settled protocol: ~ 50 ms of the same launches first (scripts/r3_clock_settle.py), then ~ 70 ms measured
    for _ in range(180): fn()
    ca.synchronize(); e0, e1 = ca.Event(), ca.Event(); e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize(); return e0.elapsed_ms(e1) / reps * 1e3
def name(c):
    return "product" if c == 0 else "GA%d GB%d halo %s%s" % (c & 7, (c >> 3) & 15, "end" if c & 128 else "start", (" no setprio" if c & 256 else "") + (" x pass through LDS" if c & 512 else ""))
cfgs = [0, 4 | (12 << 3) | 512, 4 | (10 << 3) | 512, 4 | (8 << 3) | 512]
ref = None
for c in cfgs:
    lib.mi_debug_set_long_cfg(c)
    ndi.gaussian_filter(xd, 2.0, output=o)
    got = o.get()
    if ref is None: ref = got
    print("%-26s kernel %s  equal to product: %s" % (name(c), ca.last_kernel() if hasattr(ca, "last_kernel") else "", np.array_equal(got, ref)), flush=True)
res = {c: [] for c in cfgs}
for rep in range(4):
    for c in cfgs:
        lib.mi_debug_set_long_cfg(c)
        res[c].append(t(lambda: ndi.gaussian_filter(xd, 2.0, output=o)))
        time.sleep(0.2)
for c in cfgs:
    print("%-26s %s us  (median %.1f)" % (name(c), " ".join("%.1f" % v for v in res[c]), float(np.median(res[c]))), flush=True)
lib.mi_debug_set_long_cfg(0)
