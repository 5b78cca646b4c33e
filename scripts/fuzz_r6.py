"""r6: targeted differential fuzz of this round's routes against scipy.ndimage --
  * bit-packed binary morphology (bitmorph3_kernel): erosion / dilation with 1 .. 9 iterations or until stable, masks, border
    values, origins, random structures up to 7 x 7 x 9, opening / closing (one launch up to two iterations), propagation,
    fill_holes; volumes with rows that are a multiple of 16 bytes (one and several x tiles) and images;
  * dense 3^3 / 5^3 / 7^3 correlate / convolve (stencil3s_kernel): default mode (bit-exact), float32-valued weights (v_fma_f64,
    bit-exact), dtype_mode="float" (1e-6), modes, z / y origins;
  * flat min / max with cubic sizes 3 / 5 / 7 on rows that are not a multiple of four floats (bit-exact).
usage: python scripts/fuzz_r6.py [seconds] [seed]  -> profiles/r6_fuzz_summary.txt"""
import ctypes, os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.ndimage as sndi
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
lib = _lib.load()
lib.mi_debug_set_bitmorph.argtypes = [ctypes.c_int] * 3
FMODES = ["reflect", "constant", "nearest", "mirror", "wrap"]
kernels = collections.Counter()
fails = []
cases = 0


def note():
    kernels[last_kernel()[4:].split(" ")[0].split("(")[0]] += 1


def exact(name, got, ref, info):
    global cases
    cases += 1
    note()
    if got.shape != ref.shape or not np.array_equal(got, ref):
        fails.append((name, int((np.asarray(got) != np.asarray(ref)).sum()) if got.shape == ref.shape else "shape", info))


def close(name, got, ref, tol, info):
    global cases
    cases += 1
    note()
    err = float(np.abs(got.astype(np.float64) - ref).max()) / max(float(np.abs(ref).max()), 1e-30)
    if not (err <= tol):
        fails.append((name, err, info))


def rstruct(nd):
    kind = int(rng.integers(0, 7))
    if kind == 0:
        return None
    if kind == 6 and nd == 3:                      # r-fold Minkowski sums of a 3 x 3 x 3 structure: run as iterations of the root
        r = int(rng.integers(2, 4))
        return np.ones((2 * r + 1,) * 3, bool) if rng.random() < 0.5 else np.abs(np.indices((2 * r + 1,) * 3) - r).sum(0) <= r
    if kind == 1:
        return sndi.generate_binary_structure(nd, int(rng.integers(1, nd + 1)))
    if kind == 2:
        return np.ones((3,) * nd, bool)
    shape = tuple(int(rng.integers(1, 8)) for _ in range(nd - 1)) + (int(rng.integers(1, 10)),)
    st = rng.random(shape) > rng.uniform(0.2, 0.7)
    if not st.any():
        st.flat[int(rng.integers(0, st.size))] = True
    return st


t_end = time.time() + budget
while time.time() < t_end:
    op = int(rng.integers(0, 10))
    try:
        if op <= 4:
            # ---- binary morphology
            nd = 3 if rng.random() < 0.8 else 2
            nx = int(rng.choice([64, 80, 96, 128, 176, 256, 512, 1040, 2064, 65, 71, 90, 181, 183, 255, 301, 1043, 2070]))      # r6: rows of any length
            if nd == 3:
                shape = (int(rng.integers(3, 60)), int(rng.integers(3, 90)), nx)
            else:
                shape = (int(rng.integers(40, 700)), nx)
            lib.mi_debug_set_bitmorph(2, int(rng.choice([0, 0, 0, 3, 5, 9])), int(rng.choice([0, 0, 1, 2, 4])))
            x = rng.random(shape) > rng.uniform(0.1, 0.7)
            if rng.random() < 0.2:
                x = (x * rng.integers(1, 255, size=shape)).astype(np.uint8)
            xd = ca.asarray(x)
            st = rstruct(nd)
            kw = {}
            if rng.random() < 0.4:
                kw["border_value"] = 1
            if rng.random() < 0.35:
                m = rng.random(shape) > 0.3
                kw["mask"] = m
            sshape = (3,) * nd if st is None else st.shape
            if op == 0 or op == 1:
                kw["iterations"] = int(rng.choice([1, 1, 2, 3, 4, 5, 7, 9]))
                if rng.random() < 0.3:
                    kw["origin"] = tuple(int(rng.integers(-(n // 2), (n - 1) // 2 + 1)) if n > 1 else 0 for n in sshape)
                fn, sfn = ((ndi.binary_erosion, sndi.binary_erosion), (ndi.binary_dilation, sndi.binary_dilation))[op]
            elif op == 2:
                kw["iterations"] = int(rng.choice([1, 1, 2, 3]))
                fn, sfn = ((ndi.binary_opening, sndi.binary_opening), (ndi.binary_closing, sndi.binary_closing))[int(rng.integers(0, 2))]
            elif op == 3:
                kw["iterations"] = -1
                if st is not None and not st[tuple(n // 2 for n in st.shape)]:
                    st = None                              # a structure without its centre may never become stable
                fn, sfn = ((ndi.binary_erosion, sndi.binary_erosion), (ndi.binary_dilation, sndi.binary_dilation))[int(rng.integers(0, 2))]
            else:
                kw.pop("mask", None); kw.pop("border_value", None)
                fn, sfn = ndi.binary_fill_holes, sndi.binary_fill_holes
                st = None if rng.random() < 0.5 else sndi.generate_binary_structure(nd, int(rng.integers(1, nd + 1)))   # SciPy's own until-stable path segfaults on some random structures
            kg = dict(kw)
            if "mask" in kg:
                kg["mask"] = ca.asarray(kg["mask"])
            ks = dict(kw)
            if fn is not ndi.binary_fill_holes and kw.get("iterations", 1) != 1:
                ks["brute_force"] = True          # SciPy's coordinate-list path (iterations > 1) corrupts its heap with some origins
            exact(fn.__name__, fn(xd, st, **kg).get(), sfn(x, st, **ks), (shape, None if st is None else st.shape, {k: v for k, v in kw.items() if k != "mask"}, "mask" in kw))
        elif op <= 7:
            # ---- dense 3^3 / 5^3 / 7^3 correlate (7^3: the scatter kernel in float mode, the ring kernel otherwise)
            W = int(rng.choice([3, 5, 7]))
            shape = (int(rng.integers(W, 70)), int(rng.integers(W, 90)), int(rng.choice([64, 72, 128, 181, 183, 253, 256, 258, 264, 301, 520, 1032])))       # (rows of any length since the late round)
            if np.prod(shape) < (1 << 16):
                shape = (shape[0] + 40, shape[1] + 40, shape[2])
            x = (rng.standard_normal(shape) * rng.choice([1.0, 1e3])).astype(np.float32)
            xd = ca.asarray(x)
            w = rng.standard_normal((W, W, W))
            if rng.random() < 0.4:
                w = w.astype(np.float32)
            mode = str(rng.choice(FMODES))
            org = (int(rng.integers(-(W // 2), W // 2 + 1)), int(rng.integers(-(W // 2), W // 2 + 1)), 0) if rng.random() < 0.4 else 0
            fn, sfn = ((ndi.correlate, sndi.correlate), (ndi.convolve, sndi.convolve))[int(rng.integers(0, 2))]
            ref = sfn(x.astype(np.float64), np.asarray(w, np.float64), mode=mode, cval=0.5, origin=org)
            if op == 7:
                close(fn.__name__ + "/float", fn(xd, w, mode=mode, cval=0.5, origin=org, dtype_mode="float").get(), ref, 2e-6 if W == 7 else 1e-6, (shape, W, mode, org))    # 343 float32 terms: see tests/test_gpu_stencil_scatter.py
            else:
                exact(fn.__name__, fn(xd, w, mode=mode, cval=0.5, origin=org).get(), ref.astype(np.float32), (shape, W, mode, org, str(w.dtype)))
        else:
            # ---- flat min / max on ragged rows
            nx = int(rng.choice([17, 19, 66, 181, 183, 253, 255, 257, 301, 511, 514]))
            shape = (int(rng.integers(1, 60)), int(rng.integers(3, 70)), nx)
            u = rng.random()
            if u < 0.35:                          # uint8: mm3u8_ragged_kernel (rows as they lie) from 2^15 voxels, other routes below
                x = rng.integers(0, 256, size=shape).astype(np.uint8)
            elif u < 0.6:                         # int16 / uint16: mm3s16_ragged_kernel
                x = rng.integers(-3000, 3000, size=shape).astype(np.int16) if u < 0.5 else rng.integers(0, 65536, size=shape).astype(np.uint16)
            else:
                x = rng.standard_normal(shape).astype(np.float32)
            xd = ca.asarray(x)
            size = int(rng.choice([3, 5, 7]))
            mode = str(rng.choice(FMODES))
            fn, sfn = ((ndi.minimum_filter, sndi.minimum_filter), (ndi.maximum_filter, sndi.maximum_filter), (ndi.grey_erosion, sndi.grey_erosion),
                       (ndi.grey_dilation, sndi.grey_dilation))[int(rng.integers(0, 4))]
            cv = -0.25 if x.dtype == np.float32 else 9
            exact(fn.__name__, fn(xd, size=size, mode=mode, cval=cv).get(), sfn(x, size=size, mode=mode, cval=cv), (shape, size, mode, str(x.dtype)))
    except Exception as exc:      # a refusal that is not mirrored by SciPy, or a crash: a failure either way
        fails.append(("exception", repr(exc)[:200], op))
    ca.free_all_blocks() if cases % 50 == 0 else None

lib.mi_debug_set_bitmorph(1, 0, 0)
print("r6 targeted fuzz: seed %d, %d cases in %.0f s, %d failures" % (seed, cases, budget, len(fails)))
for f in fails[:30]:
    print("  FAIL", f)
for k, n in kernels.most_common(40):
    print("   %5d  %s" % (n, k))
sys.exit(1 if fails else 0)
