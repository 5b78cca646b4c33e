"""Round-3 exploration sweep on the GPU box (not part of the product): copy ceilings, lean-kernel knobs and ablations,
long-kernel chunking, sustained (60 launches) vs first-burst timings.   python scripts/r3_explore.py [what ...]"""
import ctypes
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi

what = set(sys.argv[1:]) or {"copy", "lean", "long"}
lib = _lib.load()
n = 512
x = np.random.default_rng(0).standard_normal((n, n, n), dtype=np.float32)
xd = ca.asarray(x)
out = ca.empty(xd.shape, np.float32)
GB = 8 * xd.size / 1e9


def timeit(fn, reps=60, warm=5):
    for _ in range(warm):
        fn()
    ca.synchronize()
    e0, e1, e2 = ca.Event(), ca.Event(), ca.Event()
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    for _ in range(reps - 5):
        fn()
    e2.record()
    ca.synchronize()
    return e0.elapsed_ms(e1) / 5 * 1e3, e0.elapsed_ms(e2) / reps * 1e3        # first 5, all


def show(tag, t):
    print("%-58s first5 %7.1f us  sustained %7.1f us  %6.0f GB/s (%.3f of 8 TB/s)" % (tag, t[0], t[1], GB / t[1] * 1e6, GB / t[1] * 1e6 / 8000), flush=True)


if "copy" in what:
    lib.mi_debug_copy_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    for blocks in [256, 512, 1024, 2048, 4096, 8192, 16384, 65536]:
        show("copy_f4 blocks=%d" % blocks, timeit(lambda: lib.mi_debug_copy_f32(xd.ptr, out.ptr, xd.size, blocks, None)))
    def mc():
        out[...] = xd
    show("hipMemcpy d2d", timeit(mc))
    time.sleep(1)

if "lean" in what:
    ref = None
    for size in (5, 3, 7):
        for cfg, zch, zrev in [(0, 0, 1), (0, 0, 0), (8, 0, 1), (0, 4, 1), (0, 16, 1), (5, 4, 1), (5, 8, 1), (2, 0, 1), (1, 0, 1), (3, 0, 1)]:
            if size != 5 and cfg not in (0, 5):
                continue
            lib.mi_debug_set_sep3d_cfg(cfg); lib.mi_debug_set_sep3d_zchunks(zch); lib.mi_debug_set_sep3d_zrev(zrev)
            try:
                t = timeit(lambda: ndi.uniform_filter(xd, size=size, output=out))
            except Exception as e:
                print("size", size, "cfg", cfg, "zch", zch, "failed", repr(e)[:100]); continue
            show("lean size=%d cfg=%d zchunks=%d zrev=%d" % (size, cfg, zch, zrev), t)
            time.sleep(0.3)
        lib.mi_debug_set_sep3d_cfg(0); lib.mi_debug_set_sep3d_zchunks(0); lib.mi_debug_set_sep3d_zrev(1)
    for dbg in (1, 2, 4, 8, 3, 6, 9, 11, 15):
        lib.mi_debug_set_sep3d_dbg(dbg)
        show("lean size=5 ablation dbg=%d (1 no xz math,2 no stores,4 no loads,8 no y)" % dbg, timeit(lambda: ndi.uniform_filter(xd, size=5, output=out)))
        time.sleep(0.3)
    lib.mi_debug_set_sep3d_dbg(0)

if "long" in what:
    for sigma, zch in [(2.0, 0), (2.0, 2), (2.0, 8), (1.0, 0), (1.5, 0)]:
        lib.mi_debug_set_long_zchunks(zch)
        show("gaussian sigma=%g long zchunks=%d" % (sigma, zch), timeit(lambda: ndi.gaussian_filter(xd, sigma, output=out), reps=40))
        time.sleep(0.5)
    lib.mi_debug_set_long_zchunks(0)
    for size in (9, 13, 17):
        show("uniform size=%d" % size, timeit(lambda: ndi.uniform_filter(xd, size=size, output=out), reps=40))
        time.sleep(0.5)
