"""3 / 5 / 7-tap uniform_filter on volumes of many shapes: lean kernel (mi_debug_set_sep3d_long(1)) against the fused long kernel
(2) against the dispatch rule (0) -> profiles/r4_long_rule.txt"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit
lib = _lib.load()
rng = np.random.default_rng(0)
for shape in [(400, 500, 760), (300, 300, 300), (512, 512, 500), (256, 256, 256), (200, 1024, 1024), (512, 512, 384), (600, 600, 600), (512, 500, 512), (181, 217, 181), (512, 512, 512), (160, 384, 384), (1024, 512, 260)]:
    x = rng.standard_normal(shape).astype(np.float32); xd = ca.asarray(x); out = ca.empty(shape, np.float32)
    nv = float(np.prod(shape))
    for size in (3, 5, 7):
        row = {"shape": shape, "size": size}
        ref = None
        for knob in (1, 2, 0):
            lib.mi_debug_set_sep3d_long(knob)
            s, f = timeit(lambda: ndi.uniform_filter(xd, size, output=out), 10)
            got = out.get()
            if ref is None: ref = got
            row["knob%d" % knob] = [round(s * 1e6, 1), round(8 * nv / s / 8e12, 3), last_kernel()[4:22], float(np.abs(got - ref).max())]
        lib.mi_debug_set_sep3d_long(0)
        print(json.dumps(row), flush=True)
    del xd, out
