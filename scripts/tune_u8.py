"""config sweep of the fused uint8 min/max kernel (debug hook values 1..4; 0 = two launches)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
lib = _lib.load()
lib.mi_debug_set_u8_fused.argtypes = [ctypes.c_int]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
u = ca.asarray(np.random.default_rng(1).integers(0, 256, size=(n, n, n), dtype=np.uint8))
o = ca.empty(u.shape, np.uint8)
ref = None
for cfg in (0, 1, 2, 3, 4):
    lib.mi_debug_set_u8_fused(cfg)
    for _ in range(2): ndi.grey_erosion(u, size=7, output=o)
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(5): ndi.grey_erosion(u, size=7, output=o)
    e1.record(); ca.synchronize()
    ms = e0.elapsed_ms(e1) / 5
    got = o.get()
    if ref is None: ref = got
    print("cfg %d: %.3f ms  %.0f GB/s alg (%.1f%% of 8 TB/s)  equal=%s" % (cfg, ms, 2 * n ** 3 / ms / 1e6, 2 * n ** 3 / ms / 1e6 / 80, np.array_equal(got, ref)), flush=True)
lib.mi_debug_set_u8_fused(1)
