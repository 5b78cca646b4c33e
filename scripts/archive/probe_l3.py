"""Does a write -> read-back of a temp slab stay in the 256 MiB Infinity Cache?
Copies x -> T -> y slab by slab (T reused) for several slab sizes; 1 GiB uint8 volumes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca

n = 1024
x = ca.asarray(np.random.default_rng(0).integers(0, 255, size=(n, n, n), dtype=np.uint8))
y = ca.empty(x.shape, np.uint8)
for planes in (1024, 256, 128, 64, 32, 16):
    T = [ca.empty((planes, n, n), np.uint8) for _ in range(2)]
    def run():
        for k, z0 in enumerate(range(0, n, planes)):
            t = T[k & 1]
            t[...] = x[z0:z0 + planes]
            y[z0:z0 + planes] = t
    for _ in range(2): run()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(5): run()
    e1.record(); ca.synchronize()
    ms = e0.elapsed_ms(e1) / 5
    print("slab %4d MiB: %.3f ms for 2 x (1 GiB read + 1 GiB write)  -> %.0f GB/s of copy traffic" % (planes, ms, 4 * n ** 3 / ms / 1e6), flush=True)
    T = None
    ca.free_all_blocks()
