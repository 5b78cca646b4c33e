"""r3 / r4: anisotropic gaussians on 512^3 -- the fused long kernel with fewer z taps against the streaming passes
(mi_debug_set_sep3d_long(1)), settled protocol."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
lib = _lib.load()
n = 512
x = fs.volume_f32((n, n, n)); xd = ca.asarray(x); o = ca.empty((n, n, n), np.float32)
def t(fn):
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(10): fn()
    e1.record(); ca.synchronize()
    per = e0.elapsed_ms(e1) / 10
    for _ in range(int(40.0 / per)): fn()
    k = int(60.0 / per)
    e0.record()
    for _ in range(k): fn()
    e1.record(); ca.synchronize(); return e0.elapsed_ms(e1) / k * 1e3
for sig in ([1.0, 2.0, 2.0], [0.5, 2.0, 2.0], [0.5, 1.0, 1.0], [0.75, 1.5, 1.5],
            [2.0, 1.0, 1.0], [1.5, 1.0, 1.0], [2.0, 1.5, 1.5], [1.0, 0.5, 0.5], [1.5, 0.5, 0.5], [2.0, 0.5, 0.5], [1.5, 0.75, 0.75]):     # r4: more taps along z
    r = {}
    outs = {}
    for knob in (1, 0):
        lib.mi_debug_set_sep3d_long(knob)
        r[knob] = t(lambda: ndi.gaussian_filter(xd, sig, output=o)); outs[knob] = o.get(); name = ca.last_kernel()
    lib.mi_debug_set_sep3d_long(0)
    print("sigma %s: streaming passes %.1f us, fused %.1f us (%.3f of 8 TB/s)  max abs diff %.1e  [%s]" % (sig, r[1], r[0], 8 * n**3 / r[0] / 1e3 / 8000, float(np.abs(outs[0] - outs[1]).max()), name[:48]), flush=True)
