"""r3: lean kernel against the r3 long kernel on the slabs the strong-scaling bench gives one rank (512^3 / N planes +
halo), settled protocol; the dispatch rule must not make the multi-GPU step slower."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
lib = _lib.load()
def t(fn):
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(20): fn()
    e1.record(); ca.synchronize()
    per = e0.elapsed_ms(e1) / 20
    for _ in range(int(40.0 / per)): fn()
    n = int(60.0 / per)
    e0.record()
    for _ in range(n): fn()
    e1.record(); ca.synchronize(); return e0.elapsed_ms(e1) / n * 1e3
rng = np.random.default_rng(1)
for planes in (260, 132, 68, 36):
    x = rng.standard_normal((planes, 512, 512)).astype(np.float32); xd = ca.asarray(x); o = ca.empty(x.shape, np.float32)
    r = {}
    for knob in (1, 2, 0):
        lib.mi_debug_set_sep3d_long(knob)
        r[knob] = t(lambda: ndi.uniform_filter(xd, 5, output=o))
    lib.mi_debug_set_sep3d_long(0)
    print("%3d x 512 x 512: lean %.1f us, long3 %.1f us, auto %.1f us (%s)" % (planes, r[1], r[2], r[0], ca.last_kernel()[:40]), flush=True)
    del xd, o
