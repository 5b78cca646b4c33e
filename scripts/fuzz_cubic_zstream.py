"""Differential fuzz of the z-streaming cubic affine kernel (cubic3_zstream_kernel) against the gather kernel it replaces:
random volumes, random matrices that leave axis 0 or axis 1 to itself (rotation x shear x anisotropic scale x flips in the
(y, x) / (z, x) plane, any step along the free axis up to one plane), random offsets that push parts of the output outside, output shapes that differ
from the input's, every boundary mode, prefilter on / off.  The two kernels must agree bit for bit; every 8th case is also
checked against scipy in float64.  usage: python scripts/fuzz_cubic_zstream.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scipy.ndimage as sndi
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
lib = _lib.load()
MODES = ("constant", "nearest", "mirror", "reflect", "grid-wrap", "grid-constant", "wrap")
bad = took = 0
for i in range(cases):
    nz, ny = int(rng.integers(6, 70)), int(rng.integers(8, 200))
    nx = int(rng.integers(2, 60)) * 4
    while nz * ny * nx < (1 << 18):
        nz += 7; ny += 11
    shape = (nz, ny, nx)
    osh = shape if rng.random() < 0.5 else (int(rng.integers(4, 80)), int(rng.integers(8, 220)), int(rng.integers(64, 260)))
    while osh[0] * osh[1] * osh[2] < (1 << 18):
        osh = (osh[0] + 9, osh[1] + 13, osh[2])
    x = rng.standard_normal(shape).astype(np.float32)
    a = rng.uniform(-np.pi, np.pi) if rng.random() < 0.5 else np.deg2rad(rng.uniform(-12, 12))
    c, s = np.cos(a), np.sin(a)
    R = np.array([[c, -s], [s, c]]) @ np.array([[rng.uniform(0.6, 1.4), rng.uniform(-0.3, 0.3) * (rng.random() < 0.3)], [0, rng.uniform(0.6, 1.4)]])
    if rng.random() < 0.2:
        R[:, 1] *= -1
    M = np.eye(3); M[1:, 1:] = R
    M[0, 0] = rng.choice([1.0, -1.0, 0.5, rng.uniform(-1, 1)])
    if rng.random() < 0.4:                       # the same in the (z, x) plane: axis 1 streams
        Pm = np.array([[0, 1, 0], [1, 0, 0], [0, 0, 1.0]]); M = Pm @ M @ Pm
    off = (np.array(shape) - 1) / 2 - M @ ((np.array(osh) - 1) / 2) + rng.uniform(-6, 6, 3) * (rng.random() < 0.7)
    mode = MODES[int(rng.integers(len(MODES)))]
    kw = dict(output_shape=osh, order=3, mode=mode, cval=float(rng.uniform(-1, 1)), prefilter=bool(rng.random() < 0.6))
    xd = ca.asarray(x)
    lib.mi_debug_set_cubic_zstream(0)
    want = ndi.affine_transform(xd, M, off, **kw).get()
    lib.mi_debug_set_cubic_zstream(1 + 4 + 8 if i % 2 else 1)          # odd cases: any angle and the grid modes too
    got = ndi.affine_transform(xd, M, off, **kw).get()
    took += "cubic3_zstream_kernel" in last_kernel()
    ok = np.array_equal(got, want, equal_nan=True)
    if ok and i % 8 == 0 and kw["prefilter"]:
        ref = sndi.affine_transform(x.astype(np.float64), M, off, output_shape=osh, order=3, mode=mode, cval=kw["cval"])
        ok = np.abs(got - ref).max() <= 3e-5 * max(1.0, np.abs(ref).max())
    if not ok:
        bad += 1
        print("MISMATCH", i, shape, osh, M.tolist(), off.tolist(), kw, last_kernel()[:40], int(np.sum(got != want)), flush=True)
lib.mi_debug_set_cubic_zstream(1)
print("fuzz_cubic_zstream: %d cases (seed %d), %d took the z-streaming kernel, %d failures" % (cases, seed, took, bad))
sys.exit(1 if bad else 0)
