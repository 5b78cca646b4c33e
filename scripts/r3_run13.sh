#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
timeout 300 python bench.py --self-loop --steps 10 --warmup 3 --no-cpu > $O/bench_selfloop.json 2> $O/selfloop.err; echo "self-loop rc=$?"; cut -c1-900 $O/bench_selfloop.json; tail -3 $O/selfloop.err
timeout 600 python bench.py --config E --steps 5 --warmup 2 > $O/bench_E_n1.json 2> $O/E.err; echo "config E rc=$?"; cut -c1-1200 $O/bench_E_n1.json; tail -3 $O/E.err
timeout 300 python - <<'PY' 2>&1 | tee $O/ldpol.txt
import sys; sys.path.insert(0,'.')
import numpy as np, cupyimg_amd as ca, time
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi
lib=_lib.load()
n=512
x=ca.asarray(np.random.default_rng(0).standard_normal((n,n,n),dtype=np.float32)); o=ca.empty((n,n,n),np.float32)
def t(fn,reps=40):
    for _ in range(5): fn()
    ca.synchronize(); e0,e1=ca.Event(),ca.Event(); e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize(); return e0.elapsed_ms(e1)/reps*1e3
ref=None
for rep in range(3):
    for pol in (0,1,2,3):
        lib.mi_debug_set_sep3d_dbg(256*pol)
        u=t(lambda: ndi.uniform_filter(x,size=5,output=o))
        r=o.get()
        if ref is None: ref=r
        print("load policy %d (0 default,1 nt,2 sc1,3 sc0 sc1): %.1f us (%.3f)  same result: %s"%(pol,u,8*n**3/u/1e3/8000,bool(np.array_equal(r,ref))),flush=True)
lib.mi_debug_set_sep3d_dbg(0)
PY
