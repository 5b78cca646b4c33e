"""D' (affine order 1, SURVEY matrix, 512^3): z-stream kernel variants vs the box kernel, settled timing"""
import os, sys, json, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
from bench_configs import timeit
lib = _lib.load()
x = fs.volume_f32((512,) * 3)
xd = ca.asarray(x)
out = ca.empty(x.shape, np.float32)
M, off = fs.affine_case(512)
def run():
    ndi.affine_transform(xd, M, off, order=1, mode="constant", output=out)
res = []
for zs, zc, dbg in [(0, 0, 0), (32, 0, 0), (64, 0, 0), (32, 2, 0), (32, 8, 0), (64, 2, 0), (64, 8, 0), (64, 16, 0), (32, 0, 1), (32, 0, 2), (32, 0, 4), (32, 0, 6), (64, 0, 1), (64, 0, 2), (64, 0, 6)]:
    lib.mi_debug_set_affine_zstream(zs); lib.mi_debug_set_affine_zchunks(zc); lib.mi_debug_set_affine_dbg(dbg)
    s, f = timeit(run, 40)
    print(json.dumps({"zstream": zs, "zchunks": zc, "dbg": dbg, "us": round(s * 1e6, 1), "first_us": round(f * 1e6, 1), "frac": round(8 * 512**3 / s / 8e12, 4), "kernel": last_kernel()[:110]}), flush=True)
lib.mi_debug_set_affine_zstream(1); lib.mi_debug_set_affine_zchunks(0); lib.mi_debug_set_affine_dbg(0)
run()
print("parity whole volume:", fs.whole_volume_affine(x, M, off, out.get()))
