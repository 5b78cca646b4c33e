"""Targeted differential fuzz of the round-4 (second half) routes: (1) the z-streaming affine kernel with the sheared window against the
gather kernel, bit-exact, for random in-plane matrices (rotation x scaling x shear, flips, both streamable planes), shapes, output
shapes and offsets that push tiles over every volume edge; (2) filters on volumes / images whose rows are not a multiple of 16
bytes (the extended-rows route) against SciPy.  usage: fuzz_r4b.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi
import scipy.ndimage as sndi
lib = _lib.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t0 = time.time()
n_aff = n_aff_stream = n_rows = fails = 0
while time.time() - t0 < budget:
    # ---------------------------------------------------------------- affine
    shape = (int(rng.integers(8, 70)), int(rng.integers(8, 150)), int(rng.integers(16, 220)))
    x = rng.standard_normal(shape).astype(np.float32)
    if rng.random() < 0.3:
        x[tuple(rng.integers(0, s) for s in shape)] = np.inf
    xd = ca.asarray(x)
    a = rng.uniform(0, 2 * np.pi); c, s = np.cos(a), np.sin(a)
    A = np.array([[c, -s], [s, c]]) @ np.array([[rng.uniform(0.5, 1.3), rng.uniform(-0.5, 0.5) * (rng.random() < 0.5)], [0, rng.uniform(0.5, 1.3)]])
    ms = rng.choice([1.0, -1.0, rng.uniform(0.3, 2.0), -rng.uniform(0.3, 2.0)])
    M = np.zeros((3, 3))
    if rng.random() < 0.5:
        M[0, 0] = ms; M[1:, 1:] = A
    else:
        M[1, 1] = ms; M[0, 0], M[0, 2], M[2, 0], M[2, 2] = A[0, 0], A[0, 1], A[1, 0], A[1, 1]
    oshape = (int(rng.integers(8, 70)), int(rng.integers(8, 150)), int(rng.integers(64, 260))) if rng.random() < 0.6 else shape
    ctr_i = (np.array(shape) - 1) / 2.0; ctr_o = (np.array(oshape) - 1) / 2.0
    off = ctr_i - M @ ctr_o + rng.uniform(-20, 20, size=3) * (rng.random() < 0.7)
    cval = float(rng.uniform(-2, 2))
    lib.mi_debug_set_affine_zstream(0); lib.mi_debug_set_interp_c1(5)
    want = ndi.affine_transform(xd, M, off, output_shape=oshape, order=1, mode="constant", cval=cval).get()
    lib.mi_debug_set_interp_c1(1)
    for ty in (1, 32, 64):
        lib.mi_debug_set_affine_zstream(ty); lib.mi_debug_set_affine_zchunks(int(rng.integers(0, 4)))
        got = ndi.affine_transform(xd, M, off, output_shape=oshape, order=1, mode="constant", cval=cval).get()
        n_aff += 1; n_aff_stream += "zstream" in last_kernel()
        if not np.array_equal(got, want, equal_nan=True):
            fails += 1
            print("AFFINE MISMATCH", shape, oshape, M.tolist(), off.tolist(), ty, last_kernel()[:90], int(np.sum(~((got == want) | (np.isnan(got) & np.isnan(want))))), flush=True)
    lib.mi_debug_set_affine_zstream(1); lib.mi_debug_set_affine_zchunks(0)
    # ---------------------------------------------------------------- rows that are not a multiple of 16 bytes
    nd = 3 if rng.random() < 0.7 else 2
    shape = tuple(int(v) for v in rng.integers(12, 70, size=nd - 1)) + (int(rng.integers(33, 200)) | 1,)
    mode = str(rng.choice(["reflect", "mirror", "nearest", "wrap", "constant"]))
    cval = float(rng.integers(0, 5))
    kind = int(rng.integers(0, 6))
    dt = [np.float32, np.float32, np.float32, np.uint8, np.int16, np.float32][kind]
    x = (rng.standard_normal(shape) * 30 + 100).astype(dt)
    xd = ca.asarray(x)
    size = int(rng.choice([3, 5, 7, 9]))
    try:
        if kind == 0:
            got = ndi.uniform_filter(xd, size, mode=mode, cval=cval).get(); ref = sndi.uniform_filter(x.astype(np.float64), size, mode=mode, cval=cval); exact = False
        elif kind == 1:
            sg = float(rng.uniform(0.6, 2.4))
            got = ndi.gaussian_filter(xd, sg, mode=mode, cval=cval).get(); ref = sndi.gaussian_filter(x.astype(np.float64), sg, mode=mode, cval=cval); exact = False
        elif kind in (2, 3, 4):
            fn, rf = ((ndi.maximum_filter, sndi.maximum_filter) if rng.random() < 0.5 else (ndi.minimum_filter, sndi.minimum_filter))
            got = fn(xd, size, mode=mode, cval=cval).get(); ref = rf(x, size, mode=mode, cval=cval); exact = True
        else:
            w = rng.standard_normal((3,) * nd)
            got = ndi.correlate(xd, w, mode=mode, cval=cval).get(); ref = sndi.correlate(x, w, mode=mode, cval=cval); exact = True
        n_rows += 1
        ok = np.array_equal(got, ref) if exact else np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())
        if not ok:
            fails += 1
            print("ROWS MISMATCH", kind, shape, mode, size, dt.__name__, last_kernel()[:60], flush=True)
    except Exception as e:
        fails += 1
        print("ROWS EXCEPTION", kind, shape, mode, size, repr(e)[:200], flush=True)
print("affine cases %d (streaming kernel in %d), odd-row cases %d, failures %d" % (n_aff, n_aff_stream, n_rows, fails))
