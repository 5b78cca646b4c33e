"""r5: 3 / 5 / 7-tap separable filters on volumes whose rows are not a multiple of 16 bytes -- the fused kernel on the rows as
they are (sep3d_lean_kernel<..., ragged>) against the r4b route (mi_extend_rows + fused kernel + mi_crop_rows:
mi_debug_set_sep3d_ragged(0)).  One JSON line per row -> profiles/r5_ragged_rows.txt.   usage: python scripts/bench_ragged_rows.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit

lib = _lib.load()
rng = np.random.default_rng(0)
for shape in ((181, 217, 181), (182, 218, 184), (91, 109, 91), (193, 229, 193), (256, 256, 255), (300, 300, 301)):
    x = rng.standard_normal(shape).astype(np.float32)
    xd = ca.asarray(x)
    out = ca.empty(shape, np.float32)
    for name, fn in (("uniform_filter 3", lambda: ndi.uniform_filter(xd, 3, output=out)),
                     ("uniform_filter 5", lambda: ndi.uniform_filter(xd, 5, output=out)),
                     ("uniform_filter 7", lambda: ndi.uniform_filter(xd, 7, output=out)),
                     ("gaussian_filter 0.5 constant", lambda: ndi.gaussian_filter(xd, 0.5, mode="constant", output=out))):
        row = {"shape": shape, "call": name}
        for knob, tag in ((1, "as they are"), (0, "extended rows")):
            lib.mi_debug_set_sep3d_ragged(knob)
            try:
                t, _ = timeit(fn, 20)
            finally:
                lib.mi_debug_set_sep3d_ragged(1)
            row[tag + " us"] = round(t * 1e6, 1)
            row[tag + " kernel"] = last_kernel()[4:44]
            if knob == 1:
                row["of 8 TB/s"] = round(2 * x.nbytes / 8e12 / t, 3)
        print(json.dumps(row), flush=True)
    del xd, out
    ca.free_all_blocks()
