"""One line per API / dtype on a 256^3 volume: finds paths that are far from the HBM bound."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi

def timeit(fn, reps=3):
    fn(); ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize()
    return e0.elapsed_ms(e1) / reps

rng = np.random.default_rng(0)
n = 256
base = rng.standard_normal((n, n, n))
arrs = {"f32": base.astype(np.float32), "f64": base, "u8": (base * 40 + 128).clip(0, 255).astype(np.uint8),
        "i16": (base * 1000).astype(np.int16)}
dev = {k: ca.asarray(v) for k, v in arrs.items()}
st = np.ones((3, 3, 3), bool)
ops = [
    ("uniform_filter 5", lambda x: ndi.uniform_filter(x, 5)),
    ("gaussian_filter s=1.5", lambda x: ndi.gaussian_filter(x, 1.5)),
    ("correlate 3^3", lambda x: ndi.correlate(x, np.ones((3, 3, 3)) / 27)),
    ("minimum_filter 5", lambda x: ndi.minimum_filter(x, 5)),
    ("grey_dilation fp 3^3 cross", lambda x: ndi.grey_dilation(x, footprint=ndi.generate_binary_structure(3, 1))),
    ("median_filter 3", lambda x: ndi.median_filter(x, 3)),
    ("sobel", lambda x: ndi.sobel(x, 0)),
    ("laplace", lambda x: ndi.laplace(x)),
    ("gaussian_gradient_magnitude", lambda x: ndi.gaussian_gradient_magnitude(x, 1.0)),
    ("map_coordinates o1 (affine)", lambda x: ndi.affine_transform(x, np.eye(3) * 0.97, order=1)),
    ("affine_transform o3", lambda x: ndi.affine_transform(x, np.eye(3) * 0.97 + 0.01, order=3)),
    ("zoom 1.5 o1", lambda x: ndi.zoom(x, 1.5, order=1)),
    ("shift o3", lambda x: ndi.shift(x, 1.5, order=3)),
]
print("%-30s" % "op" + "".join("%12s" % k for k in dev))
for name, fn in ops:
    row = "%-30s" % name
    for k, d in dev.items():
        try:
            row += "%10.3fms" % timeit(lambda: fn(d))
        except Exception as exc:
            row += "%12s" % type(exc).__name__[:10]
    print(row, flush=True)
b = ca.asarray(base > 0.3)
print("binary_erosion 3^3 x1: %.3f ms; x5: %.3f ms; binary_opening: %.3f ms; fill_holes: %.3f ms" % (
    timeit(lambda: ndi.binary_erosion(b, st)), timeit(lambda: ndi.binary_erosion(b, st, iterations=5)),
    timeit(lambda: ndi.binary_opening(b, st)), timeit(lambda: ndi.binary_fill_holes(b))))
