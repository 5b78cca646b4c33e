#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vs_oracle.py -m gpu -q --maxfail=10 -k "affine or order1 or map" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python - <<'PY' 2>&1 | tee $O/interp_variants.txt
import sys, os, math
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
n=512
x=fs.volume_f32((n,n,n)); xd=ca.asarray(x); out=ca.empty(xd.shape,np.float32)
M,off=fs.affine_case(n)
def t(fn,reps=40):
    for _ in range(5): fn()
    ca.synchronize(); e0,e1=ca.Event(),ca.Event(); e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize(); return e0.elapsed_ms(e1)/reps*1e3
for v in (5,1,5,1):
    _lib.load().mi_debug_set_interp_c1(v)
    ta=t(lambda: ndi.affine_transform(xd,M,off,order=1,mode="constant",output=out))
    print("interp_c1=%d  affine %.1f us (%.3f of 8TB/s @8B)" % (v, ta, 8*n**3/ta/1e3/8000), flush=True)
def rot(axis, deg):
    a=math.radians(deg); c,s=math.cos(a),math.sin(a); R=np.eye(3); i,j=[(1,2),(0,2),(0,1)][axis]; R[i,i]=c;R[i,j]=-s;R[j,i]=s;R[j,j]=c; return R
ctr=(n-1)/2.0
for name,Mx in [("identity",np.eye(3)),("rot_x7",rot(0,7)),("rot_z5",rot(2,5)),("rot_z30",rot(2,30)),("rot_y20",rot(1,20)),("rot_y5",rot(1,5)),("zoom0.5",np.eye(3)*0.5),("zoom1.5",np.eye(3)*1.5),("zoom2",np.eye(3)*2.0)]:
    offx=ctr-Mx@np.array([ctr]*3)
    for v in (5,1):
        _lib.load().mi_debug_set_interp_c1(v)
        ta=t(lambda: ndi.affine_transform(xd,Mx,offx,order=1,mode="constant",output=out),reps=20)
        print("%-8s interp_c1=%d affine %.1f us" % (name,v,ta), flush=True)
cd=ca.asarray(fs.affine_coords_f32(n))
for v in (6,1,6,1):
    _lib.load().mi_debug_set_interp_c1(v)
    tm=t(lambda: ndi.map_coordinates(xd,cd,order=1,mode="constant",output=out))
    print("interp_c1=%d  map_coordinates %.1f us (%.3f @20B)" % (v, tm, 20*n**3/tm/1e3/8000), flush=True)
PY
timeout 200 python scripts/diag_spline_bits.py 2>&1 | head -24 | tee $O/spline_bits.txt
