"""host enqueue time per step vs device time per step for the slab schedules (self-loop)"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cupyimg_amd as ca
from cupyimg_amd.distributed import HaloComm, SlabFilter, SlabPlan, halo_widths
side, nz, size = 512, 64, 5
lo, hi = halo_widths(size)
plan = SlabPlan.self_loop(nz, lo, hi)
comm = HaloComm(1, 0, lambda u: u)
sf = SlabFilter(plan, (side, side), np.float32, comm)
sf.autotune = False
sf.ext_in.fill(1.0)
def measure(name, fn, n=300):
    for _ in range(5):
        fn(50)
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    t0 = time.perf_counter()
    fn(n)
    t1 = time.perf_counter()
    e1.record(); e1.synchronize()
    t2 = time.perf_counter()
    print(json.dumps({"what": name, "host_enqueue_us_per_step": round((t1 - t0) / n * 1e6, 2), "device_us_per_step": round(e0.elapsed_ms(e1) / n * 1e3, 2),
                      "wall_us_per_step": round((t2 - t0) / n * 1e6, 2)}))
measure("exchange only", lambda n: [comm.exchange(sf.ext_in, plan) for _ in range(n)])
measure("plain native", lambda n: [sf.uniform_filter(size, overlap=False) for _ in range(n)])
from cupyimg_amd.scipy import ndimage as ndi
measure("filter only", lambda n: [ndi.uniform_filter(sf.ext_in, size=size, output=sf.ext_out) for _ in range(n)])
for nbuf in (2, 3):
    pipe = sf.uniform_pipeline(size, nbuf=nbuf)
    for k in range(1, nbuf):
        pipe.inputs[k][...] = sf.ext_in
    measure("pipelined nbuf%d direct" % nbuf, lambda n: pipe.run(n, 0))
    measure("pipelined nbuf%d graph(1 rotation)" % nbuf, lambda n: pipe.run(n, 1), n=300)
    measure("pipelined nbuf%d graph(8 rotations)" % nbuf, lambda n: pipe.run(n, 8 * nbuf), n=8 * nbuf * 12)
    print(pipe.info())
    pipe.close()
