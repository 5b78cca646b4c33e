"""Tuning sweep for the fused separable kernel (not part of the product API).
usage: python scripts/tune_sep3d.py [n]   -- runs on the GPU box."""
import ctypes
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
size = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lib = _lib.load()
x = np.random.default_rng(0).standard_normal((n, n, n), dtype=np.float32)
xd = ca.asarray(x)
out = ca.empty(xd.shape, np.float32)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    ca.synchronize()
    return e0.elapsed_ms(e1) / reps * 1e3


lib.mi_debug_copy_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
for blocks in [512, 1024, 2048]:
    t = timeit(lambda: lib.mi_debug_copy_f32(xd.ptr, out.ptr, xd.size, blocks, None))
    print("copy blocks=%5d  %.1f us  %.0f GB/s" % (blocks, t, 8 * xd.size / t / 1e3))
ref = None
lib.mi_debug_set_sep3d_kernel(1)
ndi.uniform_filter(xd, size=size, output=out)
ref = out.get()
t = timeit(lambda: ndi.uniform_filter(xd, size=size, output=out))
print('general ws kernel: %.1f us' % t)
lib.mi_debug_set_sep3d_kernel(0)
DBG = [int(c) for c in os.environ.get('DBG', '0').split(',')]
for cfg in [int(c) for c in os.environ.get('CFGS', '0,3,10,11,12,13').split(',')]:
    for zch in [int(c) for c in os.environ.get('ZCH', '0,8,16').split(',')]:
        for dbg in DBG:
            lib.mi_debug_set_sep3d_cfg(cfg)
            lib.mi_debug_set_sep3d_zchunks(zch)
            lib.mi_debug_set_sep3d_dbg(dbg)
            try:
                t = timeit(lambda: ndi.uniform_filter(xd, size=size, output=out))
            except Exception as e:
                print("cfg", cfg, "zch", zch, "failed", e)
                continue
            o = out.get()
            if ref is None:
                ref = o
            print("cfg=%d zchunks=%2d dbg=%2d  %.1f us  %.0f GB/s (%.1f%% of 8TB/s)  maxdiff=%.2e" % (
                cfg, zch, dbg, t, 8 * xd.size / t / 1e3, 8 * xd.size / t / 1e3 / 80, float(np.abs(o - ref).max())))
