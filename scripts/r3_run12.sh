#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vs_oracle.py tests/test_gpu_baseline_full.py tests/test_gpu_fixtures.py -m gpu -q --maxfail=10 -k "affine or order1 or map or D_ or interp" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 300 python - <<'PY' 2>&1 | tee $O/interp_variants.txt
import sys, os, math
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
n=512
x=fs.volume_f32((n,n,n)); xd=ca.asarray(x); out=ca.empty(xd.shape,np.float32)
def t(fn,reps=40):
    for _ in range(5): fn()
    ca.synchronize(); e0,e1=ca.Event(),ca.Event(); e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize(); return e0.elapsed_ms(e1)/reps*1e3
cd=ca.asarray(fs.affine_coords_f32(n))
for v in (3,1,3,1,6):
    _lib.load().mi_debug_set_interp_c1(v)
    tm=t(lambda: ndi.map_coordinates(xd,cd,order=1,mode="constant",output=out))
    print("interp_c1=%d  map_coordinates %.1f us (%.3f @20B)" % (v, tm, 20*n**3/tm/1e3/8000), flush=True)
rng=np.random.default_rng(0)
wild=ca.asarray(np.stack([rng.uniform(-3,n+2,size=(128,256,256)) for _ in range(3)]).astype(np.float32))
ow=ca.empty((128,256,256),np.float32)
for v in (3,1):
    _lib.load().mi_debug_set_interp_c1(v)
    tm=t(lambda: ndi.map_coordinates(xd,wild,order=1,mode="constant",output=ow),reps=10)
    print("interp_c1=%d  map_coordinates, random coordinates, 128x256x256 outputs: %.1f us" % (v, tm), flush=True)
PY
FUZZ_ONLY=map1,affine3,zoom,shift timeout 140 python scripts/fuzz_vs_scipy.py 90 99 2>&1 | tail -3 | tee $O/fuzz.txt
FUZZ_BIG=1 FUZZ_ONLY=map1,affine3 timeout 140 python scripts/fuzz_vs_scipy.py 90 98 2>&1 | tail -3 | tee -a $O/fuzz.txt
