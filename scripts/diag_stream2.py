import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.ndimage as sndi
import cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
lib = _lib.load()
lib.mi_debug_set_stream_slice.argtypes = [ctypes.c_int]

def describe(bad, want, tag):
    m = np.argwhere(bad != want)
    print(tag, "mismatches", len(m), flush=True)
    if not len(m):
        return
    lanes = sorted(set(((m[:, 2] % 256) // 4).tolist()))
    comps = sorted(set((m[:, 2] % 4).tolist()))
    print("   lanes", lanes, "comps", comps, "z", m[:, 0].min(), m[:, 0].max(), "y", m[:, 1].min(), m[:, 1].max(), "x", m[:,2].min(), m[:,2].max())
    zs = sorted(set(m[:, 0].tolist())); ys = sorted(set(m[:, 1].tolist()))
    print("   distinct z", len(zs), zs[:20], "distinct y", len(ys), ys[:20])
    for (z, y, x) in m[:8]:
        print("   (%d,%d,%d) lane %d comp %d want %r got %r" % (z, y, x, (x % 256) // 4, x % 4, want[z, y, x], bad[z, y, x]))

rng = np.random.default_rng(190)
for shape in [(256, 256, 256), (150, 600, 64)]:
    v = rng.standard_normal(shape).astype(np.float32)
    vd = ca.asarray(v)
    for size in [(3, 1, 3), (1, 3, 3), (1, 1, 3), (3, 3, 3)]:
        lib.mi_debug_set_stream_slice(1000)
        good = ndi.minimum_filter(vd, size=size, mode="mirror").get()
        lib.mi_debug_set_stream_slice(0)
        bad = ndi.minimum_filter(vd, size=size, mode="mirror").get()
        describe(bad, good, "min %s %s" % (shape, size))
        m = np.argwhere(bad != good)
        # where did the wrong value come from?
        for (z, y, x) in m[:8]:
            loc = np.argwhere(v[max(z-2,0):z+3, max(y-2,0):y+3, max(x-40,0):x+41] == bad[z, y, x])
            print("      got value found at offsets", [(int(a + max(z-2,0) - z), int(b + max(y-2,0) - y), int(c + max(x-40,0) - x)) for a, b, c in loc][:6])
    for sigma in (2.0, 0.8):
        lib.mi_debug_set_stream_slice(1000)
        good = ndi.gaussian_filter(vd, sigma, mode="mirror").get()
        lib.mi_debug_set_stream_slice(0)
        bad = ndi.gaussian_filter(vd, sigma, mode="mirror").get()
        describe(bad, good, "gauss %s %s" % (shape, sigma))
