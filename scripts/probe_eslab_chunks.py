import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from bench_configs import timeit
lib = _lib.load()
x = np.random.default_rng(0).standard_normal((264, 2048, 2048), dtype=np.float32)
xd = ca.asarray(x); out = ca.empty(x.shape, np.float32)
for zc in (0, 1, 2, 3, 4, 6, 8):
    lib.mi_debug_set_long_zchunks(zc)
    t, _ = timeit(lambda: ndi.uniform_filter(xd, 9, output=out), 5)
    print(zc, round(t * 1e6, 1), last_kernel()[:60], flush=True)
lib.mi_debug_set_long_zchunks(0)
