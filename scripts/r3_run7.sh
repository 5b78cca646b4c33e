#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
timeout 200 python scripts/diag_spline_bits.py 2>&1 | tee $O/spline_bits.txt
timeout 600 python - <<'PY' 2>&1 | tee $O/long2.txt
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, cupyimg_amd as ca, time
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
import scipy.ndimage as sndi
from helpers import fullsize as fs
lib=_lib.load()
# correctness of the two-row kernel on odd shapes, all modes, then full size
rng=np.random.default_rng(3)
for shape in [(40,37,64),(33,21,264),(70,40,512),(19,50,256),(9,5,16)]:
    x=rng.standard_normal(shape).astype(np.float32); xd=ca.asarray(x)
    for mode in ["reflect","constant","nearest","mirror","wrap"]:
        for size in (9,13,17):
            lib.mi_debug_set_long_rows(1); a=ndi.uniform_filter(xd,size,mode=mode,cval=0.75).get()
            lib.mi_debug_set_long_rows(2); b=ndi.uniform_filter(xd,size,mode=mode,cval=0.75).get()
            ref=sndi.uniform_filter(x.astype(np.float64),size,mode=mode,cval=0.75)
            e=np.abs(b-ref).max()/np.abs(ref).max()
            if e>1e-6 or not np.array_equal(a,b): print("MISMATCH",shape,mode,size,e,np.abs(a-b).max())
        lib.mi_debug_set_long_rows(2)
        g=ndi.gaussian_filter(xd,[2.0,1.7,1.9],mode=mode,cval=-0.5).get() if False else ndi.gaussian_filter(xd,2.0,mode=mode,cval=-0.5).get()
        ref=sndi.gaussian_filter(x.astype(np.float64),2.0,mode=mode,cval=-0.5)
        e=np.abs(g-ref).max()/np.abs(ref).max()
        if e>1e-6: print("MISMATCH gaussian",shape,mode,e)
print("two-row kernel: small-shape checks done")
n=512
x=fs.volume_f32((n,n,n)); xd=ca.asarray(x); o=ca.empty((n,n,n),np.float32)
def t(fn,reps=40):
    for _ in range(5): fn()
    ca.synchronize(); e0,e1,e2=ca.Event(),ca.Event(),ca.Event(); e0.record()
    for _ in range(5): fn()
    e1.record()
    for _ in range(reps-5): fn()
    e2.record(); ca.synchronize(); return e0.elapsed_ms(e1)/5*1e3, e0.elapsed_ms(e2)/reps*1e3
for rows in (1,2,1,2):
    lib.mi_debug_set_long_rows(rows)
    for sigma in (2.0,1.5,1.0):
        a,b=t(lambda: ndi.gaussian_filter(xd,sigma,output=o))
        print("rows/wave=%d gaussian sigma=%g: first5 %.1f us sustained %.1f us (%.3f of 8 TB/s)"%(rows,sigma,a,b,8*n**3/b/1e3/8000),flush=True)
        time.sleep(0.3)
lib.mi_debug_set_long_rows(2)
ndi.gaussian_filter(xd,2.0,output=o)
print("full-size parity B (two rows):", fs.check_filter_slabs(x,o,8,8,lambda s: sndi.gaussian_filter(s.astype(np.float64),sigma=2),fs.z_slabs(n,extra=(128,256,384))))
for dbg in (0,1,2,4,8,16,7,24):
    lib.mi_debug_set_long_dbg(dbg)
    a,b=t(lambda: ndi.gaussian_filter(xd,2.0,output=o))
    print("two rows, sigma=2, dbg=%2d: first5 %.1f sustained %.1f"%(dbg,a,b),flush=True)
lib.mi_debug_set_long_dbg(0)
del xd,o; ca.free_all_blocks()
xe=fs.slab_volume_f32(fs.E_SLAB); ed=ca.asarray(xe); eo=ca.empty(fs.E_SLAB,np.float32)
for rows in (1,2,1,2):
    lib.mi_debug_set_long_rows(rows)
    a,b=t(lambda: ndi.uniform_filter(ed,size=9,output=eo),reps=15)
    print("E-slab rows/wave=%d: first5 %.1f us sustained %.1f us (%.3f)"%(rows,a,b,8*np.prod(fs.E_SLAB)/b/1e3/8000),flush=True)
PY
