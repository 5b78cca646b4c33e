#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_vs_oracle.py -m gpu -x -q -k "order1 or affine or map or interp" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 300 python - <<'PY' 2>&1 | tee $O/interp_variants.txt
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
n=512
x=fs.volume_f32((n,n,n)); xd=ca.asarray(x); out=ca.empty(xd.shape,np.float32)
M,off=fs.affine_case(n)
cd=ca.asarray(fs.affine_coords_f32(n))
def t(fn,reps=40):
    for _ in range(5): fn()
    ca.synchronize(); e0,e1=ca.Event(),ca.Event(); e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize(); return e0.elapsed_ms(e1)/reps*1e3
for v in (0,1,2,3,1,3):
    _lib.load().mi_debug_set_interp_c1(v)
    ta=t(lambda: ndi.affine_transform(xd,M,off,order=1,mode="constant",output=out))
    tm=t(lambda: ndi.map_coordinates(xd,cd,order=1,mode="constant",output=out))
    print("interp_c1=%d  affine %.1f us (%.3f of 8TB/s @8B)   map_coordinates %.1f us (%.3f @20B)" % (v, ta, 8*n**3/ta/1e3/8000, tm, 20*n**3/tm/1e3/8000), flush=True)
# other warps: rotation about z by 30 degrees, about y by 20 degrees, zoom 0.5
import math
def rot(axis, deg):
    a=math.radians(deg); c,s=math.cos(a),math.sin(a); R=np.eye(3); i,j=[(1,2),(0,2),(0,1)][axis]; R[i,i]=c;R[i,j]=-s;R[j,i]=s;R[j,j]=c; return R
ctr=(n-1)/2.0
for name,Mx in [("rot_x7",rot(0,7)),("rot_z30",rot(2,30)),("rot_y20",rot(1,20)),("zoom0.5",np.eye(3)*0.5),("zoom2",np.eye(3)*2.0)]:
    offx=ctr-Mx@np.array([ctr]*3)
    for v in (0,1,3):
        _lib.load().mi_debug_set_interp_c1(v)
        ta=t(lambda: ndi.affine_transform(xd,Mx,offx,order=1,mode="constant",output=out),reps=20)
        print("%-8s interp_c1=%d affine %.1f us" % (name,v,ta), flush=True)
PY
timeout 300 python - <<'PY' 2>&1 | tee $O/long_ablation.txt
import sys; sys.path.insert(0,'.')
import numpy as np, cupyimg_amd as ca, time
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
lib=_lib.load()
n=512
x=ca.asarray(np.random.default_rng(0).standard_normal((n,n,n),dtype=np.float32)); o=ca.empty((n,n,n),np.float32)
def t(fn,reps=40):
    for _ in range(5): fn()
    ca.synchronize(); e0,e1,e2=ca.Event(),ca.Event(),ca.Event(); e0.record()
    for _ in range(5): fn()
    e1.record()
    for _ in range(reps-5): fn()
    e2.record(); ca.synchronize(); return e0.elapsed_ms(e1)/5*1e3, e0.elapsed_ms(e2)/reps*1e3
for sigma in (2.0, 1.0):
    for dbg in (0,1,2,4,8,16,32,1|2|4,8|16,1|2|4|32,1|2|4|8|16|32, 2|4, 1|32, 0):
        lib.mi_debug_set_long_dbg(dbg)
        a,b=t(lambda: ndi.gaussian_filter(x,sigma,output=o))
        print("gaussian sigma=%g long dbg=%2d (1 y one row,2 no x,4 no z,8 no DMA,16 no stores,32 no halo tab): first5 %.1f us sustained %.1f us" % (sigma,dbg,a,b), flush=True)
        time.sleep(0.5)
lib.mi_debug_set_long_dbg(0)
PY
timeout 400 python - <<'PY' 2>&1 | tee $O/h_experiments.txt
import sys; sys.path.insert(0,'.')
import numpy as np, cupyimg_amd as ca, time
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
lib=_lib.load()
def t(fn,reps=40):
    for _ in range(5): fn()
    ca.synchronize(); e0,e1=ca.Event(),ca.Event(); e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize(); return e0.elapsed_ms(e1)/reps*1e3
for shape in [(512,512,512),(520,512,512),(504,512,512),(512,512,520),(512,520,512),(576,512,512)]:
    x=ca.asarray(np.random.default_rng(0).standard_normal(shape,dtype=np.float32)); o=ca.empty(shape,np.float32)
    res=[]
    for rep in range(3):
        for zr in (1,0):
            lib.mi_debug_set_sep3d_zrev(zr)
            res.append((zr, t(lambda: ndi.uniform_filter(x,size=5,output=o))))
    vox=np.prod(shape)
    print(shape, " ".join("zrev%d:%.1fus(%.3f)"%(z,u,8*vox/u/1e3/8000) for z,u in res), flush=True)
    del x,o; ca.free_all_blocks()
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
VAR=1 REPS=3 timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum -d $R/$O/ta1 -o s -- python3 $R/scripts/prof_interp.py > $R/$O/ta1.log 2>&1; echo "ta1 rc=$?"
VAR=1 REPS=3 timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum -d $R/$O/tcp1 -o s -- python3 $R/scripts/prof_interp.py > $R/$O/tcp1.log 2>&1; echo "tcp1 rc=$?"
cd $R/$O && python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('t*1/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'affine' in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:34],r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in sorted(agg.items()):
        print(f.split('/')[0], k, c, "%.5g"%(sum(v)/len(v)), len(v))
PY
