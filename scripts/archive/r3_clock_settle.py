"""r3: how long the clocks of a box take to settle under one kernel -- per-launch time (mean of consecutive groups of 10
launches, HIP events) over the first ~300 ms of back-to-back launches after a second of idling, for configs B, H and D'."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
n = 512
x = fs.volume_f32((n, n, n)); xd = ca.asarray(x); o = ca.empty((n, n, n), np.float32)
M, off = fs.affine_case(n)
cases = [("B gaussian sigma=2", lambda: ndi.gaussian_filter(xd, 2.0, output=o)),
         ("H uniform 5", lambda: ndi.uniform_filter(xd, 5, output=o)),
         ("D' affine order 1", lambda: ndi.affine_transform(xd, M, off, order=1, mode="constant", output=o))]
for name, fn in cases:
    fn(); ca.synchronize()
    time.sleep(1.0)
    groups = 90
    evs = [ca.Event() for _ in range(groups + 1)]
    evs[0].record()
    for g in range(groups):
        for _ in range(10):
            fn()
        evs[g + 1].record()
    ca.synchronize()
    us = [evs[g].elapsed_ms(evs[g + 1]) / 10 * 1e3 for g in range(groups)]
    t = np.cumsum([u * 10 / 1e3 for u in us])
    print(name)
    print("  ms since start :", " ".join("%6.0f" % v for v in t[::3]))
    print("  us per launch  :", " ".join("%6.1f" % v for v in us[::3]), flush=True)
