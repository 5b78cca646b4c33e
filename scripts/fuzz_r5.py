"""r5: targeted differential fuzz of the default-order (3) routes against scipy.ndimage on volumes large enough to take the round-5
kernels (one-sweep prefilter, cubic3_zfactor / zfix, row-blend with an unfiltered axis, the resampling passes, the LDS box affine
on cube tiles), of the ragged-row build of the 3 / 5 / 7-tap filter kernel and of the 128-sample rank network: random shapes, matrices, modes, output shapes.  float32 in / out: 2e-5 max(1, max|ref|); float64 prefilter: 1e-11.
usage: python scripts/fuzz_r5.py [seconds] [seed]  -> profiles/r5_fuzz_summary.txt"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.ndimage as sndi
import cupyimg_amd as ca
from cupyimg_amd import last_kernel
from cupyimg_amd.scipy import ndimage as ndi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
MODES = ["constant", "mirror", "nearest", "reflect", "grid-wrap", "wrap", "grid-constant"]
kernels = collections.Counter()
fails = []
cases = 0


def rshape():
    return (int(rng.integers(64, 150)), int(rng.integers(64, 170)), int(rng.choice([64, 68, 96, 100, 128, 132, 192, 256, 260])))


def rot(axis, deg):
    a = np.deg2rad(deg); u = np.asarray(axis, float); u /= np.linalg.norm(u)
    K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)


def check(name, got, ref, tol, info):
    global cases
    cases += 1
    kernels[last_kernel()[4:].split(" ")[0].split("(")[0]] += 1
    if got.shape != ref.shape:
        fails.append((name, "shape", got.shape, ref.shape, info)); return
    err = float(np.abs(got.astype(np.float64) - ref).max()) / max(1.0, float(np.abs(ref).max()))
    if not (err <= tol):
        fails.append((name, err, info))


t_end = time.time() + budget
while time.time() < t_end:
    shape = rshape()
    v = rng.standard_normal(shape).astype(np.float32)
    vd = ca.asarray(v)
    v64 = v.astype(np.float64)
    mode = str(rng.choice(MODES))
    op = int(rng.integers(0, 12))
    try:
        if op == 0:
            order = int(rng.choice([2, 3, 3]))
            m = str(rng.choice(["mirror", "reflect", "constant", "nearest"]))
            check("spline_filter64", ndi.spline_filter(vd, order, mode=m).get(), sndi.spline_filter(v64, order, mode=m), 1e-11, (shape, order, m))
            check("spline_filter32", ndi.spline_filter(vd, order, output=np.float32, mode=m).get(), sndi.spline_filter(v64, order, mode=m), 2e-6, (shape, order, m))
        elif op == 1:
            axes = [(1, 0), (2, 1), (2, 0)][int(rng.integers(0, 3))]
            ang = float(rng.uniform(-180, 180)); reshape = bool(rng.integers(0, 2))
            check("rotate", ndi.rotate(vd, ang, axes=axes, reshape=reshape, mode=mode, cval=0.3).get(),
                  sndi.rotate(v64, ang, axes=axes, reshape=reshape, mode=mode, cval=0.3), 2e-5, (shape, axes, ang, reshape, mode))
        elif op == 2:
            z = [float(rng.uniform(0.6, 1.8)) for _ in range(3)]
            check("zoom", ndi.zoom(vd, z, mode=mode, cval=0.3).get(), sndi.zoom(v64, z, mode=mode, cval=0.3), 2e-5, (shape, z, mode))
        elif op == 3:
            sh = [float(rng.uniform(-20, 20)) for _ in range(3)]
            check("shift", ndi.shift(vd, sh, mode=mode, cval=0.3).get(), sndi.shift(v64, sh, mode=mode, cval=0.3), 2e-5, (shape, sh, mode))
        elif op in (4, 5):
            # in-plane rotation + step along the stream axis (zfactor / zfix), random output shape
            pl = [(1, 2), (0, 2)][int(rng.integers(0, 2))]
            a = np.deg2rad(rng.uniform(-80, 80)); c, s_ = np.cos(a), np.sin(a)
            M = np.eye(3); i, j = pl
            M[i, i], M[i, j], M[j, i], M[j, j] = c, -s_, s_, c
            k = 3 - i - j
            M[k, k] = float(rng.choice([1.0, 1.02, 0.97, -1.0, 0.5, 1.25]))
            osh = tuple(int(n + rng.integers(-10, 20)) for n in shape)
            off = (np.array(shape) - 1) / 2 - M @ ((np.array(osh) - 1) / 2) + rng.uniform(-3, 3, 3)
            if rng.random() < 0.3:
                off[k] = np.round(off[k])
            check("affine3-plane", ndi.affine_transform(vd, M, off, output_shape=osh, mode=mode, cval=0.3).get(),
                  sndi.affine_transform(v64, M, off, output_shape=osh, mode=mode, cval=0.3), 2e-5, (shape, osh, pl, float(np.rad2deg(a)), M[k, k], mode))
        elif op == 6:
            M = rot(rng.standard_normal(3), float(rng.uniform(-30, 30))); off = (np.array(shape) - 1) / 2 - M @ ((np.array(shape) - 1) / 2) + rng.uniform(-3, 3, 3)
            order = int(rng.choice([1, 3]))
            check("affine-general-o%d" % order, ndi.affine_transform(vd, M, off, order=order, mode="constant", cval=0.3).get(),
                  sndi.affine_transform(v64, M, off, order=order, mode="constant", cval=0.3), 2e-5 if order == 3 else 4e-6, (shape, order))
        elif op == 8:
            # map_coordinates with its default order: smooth warp + optional jitter, float32 / float64 coordinates
            osh = tuple(int(n + rng.integers(-10, 20)) for n in shape)
            idx = np.indices(osh, dtype=np.float64)
            M = rot(rng.standard_normal(3), float(rng.uniform(-25, 25)))
            co = np.tensordot(M, idx, axes=1) + ((np.array(shape) - 1) / 2 - M @ ((np.array(osh) - 1) / 2) + rng.uniform(-3, 3, 3))[:, None, None, None]
            co += float(rng.uniform(0, 3)) * np.sin(idx[::-1] / float(rng.uniform(6, 20)))
            if rng.random() < 0.3:
                co += rng.uniform(-2, 2, co.shape)
            co = co.astype(np.float32 if rng.random() < 0.6 else np.float64)
            check("map_coordinates3", ndi.map_coordinates(vd, ca.asarray(co), mode=mode, cval=0.3).get(),
                  sndi.map_coordinates(v64, co.astype(np.float64), mode=mode, cval=0.3), 2e-5, (shape, osh, mode, str(co.dtype)))
        elif op == 9:
            # rows that are not a multiple of four floats through the 3 / 5 / 7-tap fused kernel (ragged build)
            sh = (int(rng.integers(3, 60)), int(rng.integers(3, 80)), int(rng.choice([17, 19, 66, 101, 181, 253, 255, 257, 301, 511, 515])) + int(rng.integers(0, 3)) * 4)
            w = rng.standard_normal(sh).astype(np.float32); wd = ca.asarray(w)
            m = str(rng.choice(["reflect", "mirror", "nearest", "wrap", "constant"]))
            if rng.random() < 0.5:
                size = int(rng.choice([3, 5, 7]))
                check("uniform-ragged", ndi.uniform_filter(wd, size, mode=m, cval=0.3).get(), sndi.uniform_filter(w.astype(np.float64), size, mode=m, cval=0.3), 1e-6, (sh, size, m))
            else:
                sg = float(rng.choice([0.25, 0.4, 0.5, 0.6, 0.75, 0.8]))
                check("gaussian-ragged", ndi.gaussian_filter(wd, sg, mode=m, cval=0.3).get(), sndi.gaussian_filter(w.astype(np.float64), sg, mode=m, cval=0.3), 1e-6, (sh, sg, m))
        elif op == 10:
            # rank filters with 65 .. 128 samples (sorting network on 128 registers): bit-exact
            dt = [np.float32, np.uint8, np.int16, np.uint16, np.int8][int(rng.integers(0, 5))]
            nd = int(rng.choice([2, 3]))
            sh = (int(rng.integers(6, 24)), int(rng.integers(6, 40)), int(rng.integers(8, 140)))[3 - nd:]
            w = (rng.standard_normal(sh) * 50).astype(dt); wd = ca.asarray(w)
            fshape = tuple(int(rng.integers(3, 7)) for _ in range(nd)) if nd == 3 else tuple(int(rng.integers(7, 13)) for _ in range(nd))
            fp = rng.random(fshape) < float(rng.uniform(0.5, 1.0))
            nset = int(fp.sum())
            if not (65 <= nset <= 128):
                continue
            rank = int(rng.integers(0, nset))
            m = str(rng.choice(["reflect", "mirror", "nearest", "wrap", "constant"]))
            org = tuple(int(rng.integers(-(f // 2), f // 2 + (f & 1))) if rng.random() < 0.3 else 0 for f in fshape)
            check("rank-128", ndi.rank_filter(wd, rank, footprint=fp, mode=m, cval=3, origin=org).get(), sndi.rank_filter(w, rank, footprint=fp, mode=m, cval=3, origin=org), 0.0,
                  (sh, str(np.dtype(dt)), fshape, nset, rank, m, org))
        elif op == 11:
            # the 3 x 3 x 3 median on the kernel that shares its sorting between windows: bit-exact
            dt = [np.float32, np.uint8, np.int16, np.uint16, np.int8, np.int32, np.uint32][int(rng.integers(0, 7))]
            sh = (int(rng.integers(2, 70)), int(rng.integers(2, 90)), int(rng.integers(8, 300)))
            if sh[0] * sh[1] * sh[2] < 4096:
                continue
            w = (rng.standard_normal(sh) * 50).astype(dt); wd = ca.asarray(w)
            m = str(rng.choice(["reflect", "mirror", "nearest", "wrap", "constant"]))
            if rng.random() < 0.5:
                check("median27", ndi.median_filter(wd, size=3, mode=m, cval=-3).get(), sndi.median_filter(w, size=3, mode=m, cval=-3), 0.0, (sh, str(np.dtype(dt)), m))
            else:
                rk = int(rng.integers(1, 26))
                check("rank27", ndi.rank_filter(wd, rk, size=3, mode=m, cval=-3).get(), sndi.rank_filter(w, rk, size=3, mode=m, cval=-3), 0.0, (sh, str(np.dtype(dt)), m, rk))
        else:
            M = rot((0, 0, 1), float(rng.uniform(-40, 40)))      # rotation in the (z, y) plane: x to itself (row-blend)
            off = (np.array(shape) - 1) / 2 - M @ ((np.array(shape) - 1) / 2)
            off[2] = float(rng.integers(-5, 6))
            check("affine3-rowblend", ndi.affine_transform(vd, M, off, mode=mode, cval=0.3).get(),
                  sndi.affine_transform(v64, M, off, mode=mode, cval=0.3), 2e-5, (shape, mode, off[2]))
    except Exception as exc:      # noqa
        fails.append(("exception", repr(exc)[:200], (shape, op, mode)))
print("r5 targeted fuzz: seed %d, %d cases in %.0f s, %d failures" % (seed, cases, budget, len(fails)))
for k, n in kernels.most_common():
    print("   %5d  %s" % (n, k))
for f in fails[:20]:
    print("FAIL", f)
sys.exit(1 if fails else 0)
