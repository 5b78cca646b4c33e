"""A/B of the non-temporal hint (mi_debug_set_stream_nt 0 / -1) over volume shapes and filter sizes, same process,
interleaved, best of two settled runs -> profiles/r4_stream_nt.txt (second table)"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit
lib = _lib.load()
for shape, sizes in [((264, 2048, 2048), (3, 5, 7, 9)), ((200, 1024, 1024), (5, 9)), ((128, 768, 1536), (5, 9)), ((512, 512, 512), (5, 9))]:
    xd = ca.empty(shape, np.float32); xd.fill(1.0)
    out = ca.empty(shape, np.float32)
    vox = shape[0] * shape[1] * shape[2]
    for size in sizes:
        r = {}
        for rep in range(2):
            for nt in (0, -1):
                lib.mi_debug_set_stream_nt(nt)
                s, f = timeit(lambda: ndi.uniform_filter(xd, size, output=out), 20)
                r.setdefault(nt, []).append(s * 1e6)
        print(shape, "size", size, "no hint %.1f us (%.3f)  hint %.1f us (%.3f)  %s" % (min(r[0]), 8 * vox / min(r[0]) / 8e6, min(r[-1]), 8 * vox / min(r[-1]) / 8e6, ca.last_kernel()[:48]), flush=True)
    del xd, out
    ca.free_all_blocks()
