"""r3: the r3 long-kernel instruction stream against the lean kernel at 3 / 5 / 7 taps (mi_debug_set_sep3d_long:
1 = lean kernel, 2 = long kernel also for 3..7 taps), over a set of volume shapes; interleaved runs."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
lib = _lib.load()
def t(fn, reps):
    for _ in range(max(10, reps // 5)): fn()
    ca.synchronize(); e0, e1 = ca.Event(), ca.Event(); e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize(); return e0.elapsed_ms(e1) / reps * 1e3
rng = np.random.default_rng(5)
shapes = [(512, 512, 512), (256, 256, 256), (128, 128, 128), (64, 64, 64), (300, 300, 300), (64, 1024, 1024), (1024, 128, 128),
          (200, 500, 760), (40, 37, 64), (17, 200, 2048)]
for shape in shapes:
    x = rng.standard_normal(shape).astype(np.float32); xd = ca.asarray(x); o = ca.empty(shape, np.float32)
    nvox = int(np.prod(shape))
    reps = int(min(150, max(20, 3e9 / nvox / 8)))
    for size in (3, 5, 7):
        res = {1: [], 2: []}
        outs = {}
        for rep in range(2):
            for knob in (1, 2):
                lib.mi_debug_set_sep3d_long(knob)
                res[knob].append(t(lambda: ndi.uniform_filter(xd, size, output=o), reps))
                if rep == 0: outs[knob] = o.get()
        lib.mi_debug_set_sep3d_long(0)
        a, b = min(res[1]), min(res[2])
        print("%-18s size %d: lean %8.1f us  long3 %8.1f us  ratio %.3f  (%.3f of 8 TB/s)  max abs diff %.1e" % (shape, size, a, b, b / a, 8 * nvox / b / 1e3 / 8000, float(np.abs(outs[1] - outs[2]).max())), flush=True)
    del xd, o
    ca.free_all_blocks()
