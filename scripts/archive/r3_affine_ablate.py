"""r3: phase ablations of affine3d_lds_kernel on config D' (mi_debug_set_affine_dbg: 1 no box DMA, 2 no interpolation,
4 no stores) -- timing only, the results of an ablated run are meaningless."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
lib = _lib.load()
n = 512
x = fs.volume_f32((n, n, n)); xd = ca.asarray(x); out = ca.empty(xd.shape, np.float32)
M, off = fs.affine_case(n)
def t(fn, reps=40):
    for _ in range(8): fn()
    ca.synchronize(); e0, e1 = ca.Event(), ca.Event(); e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize(); return e0.elapsed_ms(e1) / reps * 1e3
f = lambda: ndi.affine_transform(xd, M, off, order=1, mode="constant", output=out)
for gz in (0, 64, 3, 6, 12, 24, 0):
    lib.mi_debug_set_affine_gz(gz)
    print("affine3d_lds workgroups along z = %d (0 auto, 64 = one tile each): %.1f us" % (gz, t(f)), flush=True)
lib.mi_debug_set_affine_gz(0)
for dbg in (0, 1, 2, 4, 3, 5, 6, 7, 0):
    lib.mi_debug_set_affine_dbg(dbg)
    print("affine3d_lds dbg=%d (1 no DMA, 2 no interpolation, 4 no stores): %.1f us" % (dbg, t(f)), flush=True)
    time.sleep(0.2)
lib.mi_debug_set_affine_dbg(0)
