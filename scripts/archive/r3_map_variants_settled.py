"""r3: the map_coordinates order-1 kernel variants (mi_debug_set_interp_c1) on config D under the settled protocol."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
lib = _lib.load()
n = 512
x = fs.volume_f32((n, n, n)); xd = ca.asarray(x); out = ca.empty(xd.shape, np.float32)
cd = ca.asarray(fs.affine_coords_f32(n))
def t(fn, reps=110):
    for _ in range(80): fn()
    ca.synchronize(); e0, e1 = ca.Event(), ca.Event(); e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize(); return e0.elapsed_ms(e1) / reps * 1e3
f = lambda: ndi.map_coordinates(xd, cd, order=1, mode="constant", output=out)
names = {0: "round-2 kernel", 1: "default (z-major c1)", 2: "c1 narrow stores", 6: "c1 row-major", 4: "LDS-staged", 7: "pair sharing"}
ref = None
for var in (1, 0, 2, 6, 4, 7, 1):
    lib.mi_debug_set_interp_c1(var)
    us = t(f)
    got = out.get()
    if ref is None: ref = got
    print("knob %d %-22s %.1f us (%.3f of 8 TB/s)  equal to default: %s" % (var, names[var], us, 20 * n**3 / us / 1e3 / 8000, np.array_equal(got, ref, equal_nan=True)), flush=True)
lib.mi_debug_set_interp_c1(1)
