"""r3: sep3d_long3_kernel (y pass one plane ahead, op_sel x pass) against the r2 kernel (mi_debug_set_long_rows(1)):
small-shape parity in every mode, full-size parity of config B, sustained timings at 17 / 13 / 9 taps and on the
E slab, ablations of the new kernel.  Run on the GPU box: python scripts/r3_long3.py [quick]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
import scipy.ndimage as sndi
from helpers import fullsize as fs
lib = _lib.load()
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
rng = np.random.default_rng(3)
bad = 0
for shape in [(40, 37, 64), (33, 21, 264), (70, 40, 512), (19, 50, 256), (9, 5, 16), (35, 33, 300)]:
    x = rng.standard_normal(shape).astype(np.float32); xd = ca.asarray(x)
    for mode in ["reflect", "constant", "nearest", "mirror", "wrap"]:
        for size in (3, 5, 7, 9, 11, 13, 15, 17):
            lib.mi_debug_set_sep3d_long(2)        # route every cubic kernel through the long path
            lib.mi_debug_set_long_rows(0)
            b = ndi.uniform_filter(xd, size, mode=mode, cval=0.75).get()
            kname = ca.last_kernel() if hasattr(ca, "last_kernel") else ""
            ref = sndi.uniform_filter(x.astype(np.float64), size, mode=mode, cval=0.75)
            e = np.abs(b - ref).max() / np.abs(ref).max()
            if e > 1e-6:
                bad += 1; print("MISMATCH uniform", shape, mode, size, e, kname)
        lib.mi_debug_set_sep3d_long(0)
        for sig in (2.0, [2.0, 1.7, 1.9], 1.5):
            g = ndi.gaussian_filter(xd, sig, mode=mode, cval=-0.5).get()
            ref = sndi.gaussian_filter(x.astype(np.float64), sig, mode=mode, cval=-0.5)
            e = np.abs(g - ref).max() / np.abs(ref).max()
            if e > 1e-6:
                bad += 1; print("MISMATCH gaussian", shape, mode, sig, e)
        # origins on y / z
        for org in ([2, -3, 0], [-4, 4, 0]):
            b = ndi.uniform_filter(xd, 11, mode=mode, origin=org).get()
            ref = sndi.uniform_filter(x.astype(np.float64), 11, mode=mode, origin=org)
            e = np.abs(b - ref).max() / np.abs(ref).max()
            if e > 1e-6:
                bad += 1; print("MISMATCH origin", shape, mode, org, e)
print("long3 kernel: small-shape checks done, mismatches:", bad, flush=True)
n = 512
x = fs.volume_f32((n, n, n)); xd = ca.asarray(x); o = ca.empty((n, n, n), np.float32)
def t(fn, reps=40):
    """(mean of the first five launches after a pause, settled mean): the clocks of a box need ~ 40 ms of load to
    settle (scripts/r3_clock_settle.py), so the settled figure is taken over >= 60 ms after >= 50 ms of the same launches"""
    ca.synchronize(); time.sleep(0.3)
    e0, e1, e2, e3 = ca.Event(), ca.Event(), ca.Event(), ca.Event(); e0.record()
    for _ in range(5): fn()
    e1.record(); ca.synchronize()
    per = e0.elapsed_ms(e1) / 5
    for _ in range(int(50.0 / per) + 1): fn()
    n2 = max(reps, int(60.0 / per))
    e2.record()
    for _ in range(n2): fn()
    e3.record(); ca.synchronize(); return per * 1e3, e2.elapsed_ms(e3) / n2 * 1e3
lib.mi_debug_set_long_rows(0)
ndi.gaussian_filter(xd, 2.0, output=o)
print("full-size parity B (long3):", fs.check_filter_slabs(x, o, 8, 8, lambda s: sndi.gaussian_filter(s.astype(np.float64), sigma=2), fs.z_slabs(n, extra=(128, 256, 384))), flush=True)
for rows in (1, 0, 1, 0):
    lib.mi_debug_set_long_rows(rows)
    for sigma in (2.0, 1.5, 1.0):
        a, b = t(lambda: ndi.gaussian_filter(xd, sigma, output=o))
        print("kernel=%s gaussian sigma=%g: first5 %.1f us settled %.1f us (%.3f of 8 TB/s)" % ("r2" if rows else "r3", sigma, a, b, 8 * n**3 / b / 1e3 / 8000), flush=True)
        time.sleep(0.3)
lib.mi_debug_set_long_rows(0)
if not quick:
    for dbg in (0, 128, 64, 128, 64, 1, 2, 4, 8, 16, 32, 7, 24, 63):
        lib.mi_debug_set_long_dbg(dbg)
        a, b = t(lambda: ndi.gaussian_filter(xd, 2.0, output=o))
        print("r3 kernel, sigma=2, dbg=%2d: first5 %.1f settled %.1f" % (dbg, a, b), flush=True)
    lib.mi_debug_set_long_dbg(0)
del xd, o; ca.free_all_blocks()
xe = fs.slab_volume_f32(fs.E_SLAB); ed = ca.asarray(xe); eo = ca.empty(fs.E_SLAB, np.float32)
for rows in (1, 0, 1, 0):
    lib.mi_debug_set_long_rows(rows)
    a, b = t(lambda: ndi.uniform_filter(ed, size=9, output=eo), reps=15)
    print("E-slab kernel=%s: first5 %.1f us settled %.1f us (%.3f)" % ("r2" if rows else "r3", a, b, 8 * np.prod(fs.E_SLAB) / b / 1e3 / 8000), flush=True)
lib.mi_debug_set_long_rows(0)
