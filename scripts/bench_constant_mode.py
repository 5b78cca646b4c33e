"""r5: `constant` mode on 512^3 float32 -- zero fill on the r3 long kernel (default), the r2 kernel with its coverage correction
(mi_debug_set_long_const0(0)), and the same filter in `reflect` mode.  -> profiles/r5_constant_mode.txt
usage: python scripts/bench_constant_mode.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit
lib = _lib.load()
x = np.random.default_rng(0).standard_normal((512, 512, 512)).astype(np.float32)
xd = ca.asarray(x); out = ca.empty(x.shape, np.float32)
for name, fn in (("gaussian_filter sigma 2 (17 taps)", lambda m, **kw: ndi.gaussian_filter(xd, 2.0, mode=m, output=out, **kw)),
                 ("gaussian_filter sigma 1.5 (13 taps)", lambda m, **kw: ndi.gaussian_filter(xd, 1.5, mode=m, output=out, **kw)),
                 ("uniform_filter 9", lambda m, **kw: ndi.uniform_filter(xd, 9, mode=m, output=out, **kw)),
                 ("uniform_filter 5", lambda m, **kw: ndi.uniform_filter(xd, 5, mode=m, output=out, **kw)),
                 ("uniform_filter 3", lambda m, **kw: ndi.uniform_filter(xd, 3, mode=m, output=out, **kw)),
                 ("gaussian_filter sigma 0.75 (7 taps)", lambda m, **kw: ndi.gaussian_filter(xd, 0.75, mode=m, output=out, **kw)),
                 ("gaussian_filter sigma (1, 2, 2) (9 / 17 / 17 taps)", lambda m, **kw: ndi.gaussian_filter(xd, (1.0, 2.0, 2.0), mode=m, output=out, **kw))):
    row = {"call": name}
    t, _ = timeit(lambda: fn("reflect"), 20); row["reflect us"] = round(t * 1e6, 1)
    t, _ = timeit(lambda: fn("constant"), 20); row["constant (cval 0) us"] = round(t * 1e6, 1); row["kernel"] = last_kernel()[4:44]
    row["of 8 TB/s"] = round(2 * x.nbytes / 8e12 / t, 3)
    lib.mi_debug_set_long_const0(0)
    t, _ = timeit(lambda: fn("constant"), 20); row["r2 kernel us"] = round(t * 1e6, 1); row["r2 kernel"] = last_kernel()[4:44]
    lib.mi_debug_set_long_const0(1)
    t, _ = timeit(lambda: fn("constant", cval=1.0), 20); row["constant (cval 1) us"] = round(t * 1e6, 1)
    print(json.dumps(row), flush=True)
