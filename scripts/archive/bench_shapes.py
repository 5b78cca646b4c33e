"""Fused uniform_filter on a few volume shapes (tile / chunk plan check).

    python scripts/bench_shapes.py [sizes, e.g. 3,5] [shape set: small|big]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi

def timeit(fn, reps=10):
    for _ in range(3): fn()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize()
    return e0.elapsed_ms(e1) / reps * 1e3

sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3,5").split(",")]
which = sys.argv[2] if len(sys.argv) > 2 else "small"
SHAPES = {
    "small": [(512, 512, 512), (68, 512, 512), (132, 512, 512), (260, 512, 512), (300, 300, 300), (100, 1000, 1024), (512, 512, 256)],
    "big": [(512, 512, 512), (264, 1024, 1024), (264, 512, 2048), (264, 2048, 512), (66, 2048, 2048), (264, 2048, 2048), (1024, 1024, 1024)],
}[which]
for shape in SHAPES:
    x = ca.asarray(np.random.default_rng(0).standard_normal(shape, dtype=np.float32))
    o = ca.empty(shape, np.float32)
    for size in sizes:
        t = timeit(lambda: ndi.uniform_filter(x, size=size, output=o))
        n = np.prod(shape)
        print("shape %-18s size %d  %8.1f us  %6.0f GB/s" % (shape, size, t, 8 * n / t / 1e3), flush=True)
    x = o = None
    ca.free_all_blocks()
