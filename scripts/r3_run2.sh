#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vs_oracle.py tests/test_gpu_halo.py tests/test_gpu_images2d.py tests/test_gpu_baseline_full.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for v in 0 1 2; do
python - <<PY 2>&1 | tee -a $O/interp_variants.txt
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
_lib.load().mi_debug_set_interp_c1($v)
n=512
x=fs.volume_f32((n,n,n)); xd=ca.asarray(x); out=ca.empty(xd.shape,np.float32)
M,off=fs.affine_case(n)
def t(fn,reps=40):
    for _ in range(5): fn()
    ca.synchronize(); e0,e1=ca.Event(),ca.Event(); e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize(); return e0.elapsed_ms(e1)/reps*1e3
ta=t(lambda: ndi.affine_transform(xd,M,off,order=1,mode="constant",output=out))
cd=ca.asarray(fs.affine_coords_f32(n))
tm=t(lambda: ndi.map_coordinates(xd,cd,order=1,mode="constant",output=out))
print("interp_c1=$v  affine %.1f us (%.3f of 8TB/s @8B)   map_coordinates %.1f us (%.3f @20B)" % (ta, 8*n**3/ta/1e6/8000, tm, 20*n**3/tm/1e6/8000))
PY
done
VAR=0 bash scripts/pmc_interp.sh r3b/aff_v0 > /dev/null
VAR=1 bash scripts/pmc_interp.sh r3b/aff_v1 > /dev/null
VAR=0 MAP=1 bash scripts/pmc_interp.sh r3b/map_v0 > /dev/null
VAR=1 MAP=1 bash scripts/pmc_interp.sh r3b/map_v1 > /dev/null
cd $GRAFT_REPO_ROOT
cat $O/aff_v0/summary.txt $O/aff_v1/summary.txt $O/map_v0/summary.txt $O/map_v1/summary.txt
python - <<'PY' 2>&1 | tee $O/h_experiments.txt
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
    print(shape, " ".join("zrev%d:%.1fus(%.3f)"%(z,u,8*vox/u/1e6/8000) for z,u in res))
    del x,o; ca.free_all_blocks()
PY
