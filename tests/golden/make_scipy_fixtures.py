"""Generates tests/golden/scipy_fixtures.npz from SciPy (the arithmetic truth
the reference's own tests compare against, cupyimg/testing/helper.py:51-68).

Run in the authoring container:   python tests/golden/make_scipy_fixtures.py
SciPy / NumPy versions are recorded in the file.  The fixture holds inputs,
arguments and SciPy's outputs only -- no reference source.

Case families follow SURVEY.md section 8c: uniform / gaussian / correlate1d
sweep (tests/test_ndimage_vs_scipy.py:15-102 of the reference) / n-D
correlate (tests/test_filters_from_cupy.py:18-94) / min-max-grey / binary
(tests/test_morphology_from_cupy.py:378-425) / interpolation
(tests/test_interpolation.py:24-243) plus one down-scaled replica of every
BASELINE.json config.
"""
import json
import os
import sys

import numpy as np
import scipy
import scipy.ndimage as ndi

HERE = os.path.dirname(os.path.abspath(__file__))
MODES = ["reflect", "constant", "nearest", "mirror", "wrap"]
IMODES = ["constant", "grid-constant", "nearest", "mirror", "reflect", "wrap", "grid-wrap"]

arrays = {}
cases = []


_seen = {}


def put(arr):
    arr = np.ascontiguousarray(arr)
    key = (arr.dtype.str, arr.shape, arr.tobytes())
    if key in _seen:
        return _seen[key]
    name = "a%d" % len(arrays)
    arrays[name] = arr
    _seen[key] = name
    return name


def case(func, arrs, kwargs, expected, tol=None, family=""):
    cases.append({
        "id": len(cases), "func": func, "family": family,
        "arrays": {k: put(v) for k, v in arrs.items()},
        "kwargs": kwargs, "expected": put(expected), "tol": tol,
    })


def rnd(rng, shape, dtype):
    dtype = np.dtype(dtype)
    if dtype.kind == "f":
        return rng.standard_normal(shape).astype(dtype)
    if dtype.kind == "b":
        return rng.random(shape) > 0.5
    info = np.iinfo(dtype)
    lo, hi = max(info.min, -100), min(info.max, 200)
    return rng.integers(lo, hi, size=shape, endpoint=True).astype(dtype)


def main():
    rng = np.random.default_rng(20261001)

    # ---------------------------------------------------------------- uniform
    for shape in [(12, 13, 16), (11, 13), (23,)]:
        for dt in ["float32", "float64", "uint8", "int16"]:
            x = rnd(rng, shape, dt)
            for size in [3, 5, 9]:
                for mode in MODES:
                    kw = dict(size=size, mode=mode, cval=1.25 if np.dtype(dt).kind == "f" else 3)
                    case("uniform_filter", {"input": x}, kw, ndi.uniform_filter(x, **kw), family="uniform")
            for origin in [-1, 1]:
                kw = dict(size=4, mode="reflect", origin=origin)
                case("uniform_filter", {"input": x}, kw, ndi.uniform_filter(x, **kw), family="uniform")
            for odt in ["float32", "float64"]:
                kw = dict(size=5, mode="nearest", output=odt)
                case("uniform_filter", {"input": x}, kw, ndi.uniform_filter(x, **kw), family="uniform")
    x = rnd(rng, (10, 12, 16), "float32")
    kw = dict(size=(3, 1, 5), mode=["reflect", "nearest", "mirror"])
    case("uniform_filter", {"input": x}, kw, ndi.uniform_filter(x, **kw), family="uniform")
    kw = dict(size=(5, 3, 7), mode=["wrap", "constant", "reflect"], cval=0.0)
    case("uniform_filter", {"input": x}, kw, ndi.uniform_filter(x, **kw), family="uniform")
    for ax in range(3):
        kw = dict(size=5, axis=ax, mode="mirror", origin=1)
        case("uniform_filter1d", {"input": x}, kw, ndi.uniform_filter1d(x, **kw), family="uniform")

    # --------------------------------------------------------------- gaussian
    for shape in [(12, 13, 16), (14, 15)]:
        for dt in ["float32", "float64", "uint8"]:
            x = rnd(rng, shape, dt)
            for sigma in [0.5, 1.0, 2.0]:
                for order in [0, 1, 2]:
                    for truncate in [2.0, 4.0]:
                        kw = dict(sigma=sigma, order=order, truncate=truncate)
                        case("gaussian_filter", {"input": x}, kw, ndi.gaussian_filter(x, **kw),
                             family="gaussian")
            for mode in MODES:
                kw = dict(sigma=1.0, mode=mode, cval=0.5)
                case("gaussian_filter", {"input": x}, kw, ndi.gaussian_filter(x, **kw), family="gaussian")
    x = rnd(rng, (9, 10, 12), "float32")
    kw = dict(sigma=(1.0, 0.0, 0.7), order=(0, 0, 1))
    case("gaussian_filter", {"input": x}, kw, ndi.gaussian_filter(x, **kw), family="gaussian")
    kw = dict(sigma=1.5, axis=1, order=1, mode="nearest")
    case("gaussian_filter1d", {"input": x}, kw, ndi.gaussian_filter1d(x, **kw), family="gaussian")

    # ------------------------------------------------- correlate1d / convolve1d
    for dt in ["float64", "int32"]:
        for n in [1, 2, 3, 6, 7]:
            x = rnd(rng, (n,), dt)
            for wl in [1, 2, 3, 4, 5, 2 * n + 1]:
                w = rng.standard_normal(wl)
                for mode in MODES:
                    for origin in range(-(wl // 2), wl - wl // 2):
                        for fn in ["correlate1d", "convolve1d"]:
                            kw = dict(mode=mode, origin=origin, cval=2.0, output="float64")
                            try:
                                exp = getattr(ndi, fn)(x, w, **kw)
                            except ValueError:
                                continue
                            case(fn, {"input": x, "weights": w}, kw, exp, family="corr1d")
    x = rnd(rng, (7, 8, 9), "float32")
    for ax in range(3):
        for w in [np.array([1.0, 2.0, 1.0]), np.array([1.0, 0.0, -1.0]), rng.standard_normal(4),
                  rng.standard_normal(70)]:
            for fn in ["correlate1d", "convolve1d"]:
                kw = dict(axis=ax, mode="reflect")
                case(fn, {"input": x, "weights": w}, kw, getattr(ndi, fn)(x, w, **kw), family="corr1d")

    # ------------------------------------------------------- n-D correlate
    for shape, wshapes in [((8, 9, 10), [(3, 3, 3), (2, 3, 4), (1, 5, 1)]), ((11, 12), [(3, 3), (4, 2), (5, 5)]),
                           ((15,), [(3,), (4,)])]:
        for dt in ["float32", "float64", "uint8", "int32"]:
            x = rnd(rng, shape, dt)
            for ws in wshapes:
                w = rng.standard_normal(ws)
                w[w < -1] = 0
                for mode in MODES:
                    for fn in ["correlate", "convolve"]:
                        kw = dict(mode=mode, cval=2.0)
                        case(fn, {"input": x, "weights": w}, kw, getattr(ndi, fn)(x, w, **kw), family="corrnd")
                org = [(-1 if s > 2 else 0) for s in ws]
                for fn in ["correlate", "convolve"]:
                    kw = dict(mode="reflect", origin=org, output="float64")
                    case(fn, {"input": x, "weights": w}, kw, getattr(ndi, fn)(x, w, **kw), family="corrnd")

    # ------------------------------------------------------------- min / max
    for shape in [(9, 10, 11), (12, 13)]:
        nd = len(shape)
        for dt in ["float32", "float64", "uint8", "int16", "int32"]:
            x = rnd(rng, shape, dt)
            for size in [2, 3, 7]:
                for mode in MODES:
                    for fn in ["minimum_filter", "maximum_filter", "grey_erosion", "grey_dilation"]:
                        kw = dict(size=size, mode=mode, cval=3)
                        case(fn, {"input": x}, kw, getattr(ndi, fn)(x, **kw), family="minmax")
            fs = (3,) * nd
            fp = rng.random(fs) > 0.4
            fp.flat[0] = True
            st = np.round(rng.standard_normal(fs) * 5)
            for mode in MODES:
                for fn in ["minimum_filter", "maximum_filter"]:
                    kw = dict(mode=mode, cval=3)
                    case(fn, {"input": x, "footprint": fp}, kw, getattr(ndi, fn)(x, footprint=fp, **kw),
                         family="minmax")
                for fn in ["grey_erosion", "grey_dilation"]:
                    kw = dict(mode=mode, cval=3)
                    case(fn, {"input": x, "footprint": fp, "structure": st}, kw,
                         getattr(ndi, fn)(x, footprint=fp, structure=st, **kw), family="minmax")
            fs = (2, 3, 4)[:nd]
            fp = np.ones(fs, bool)
            fp.flat[1] = False
            for fn in ["minimum_filter", "maximum_filter", "grey_erosion", "grey_dilation"]:
                kw = dict(mode="reflect", origin=[0, -1, 1][:nd])
                case(fn, {"input": x, "footprint": fp}, kw, getattr(ndi, fn)(x, footprint=fp, **kw),
                     family="minmax")
            for ax in range(nd):
                for fn in ["minimum_filter1d", "maximum_filter1d"]:
                    kw = dict(size=4, axis=ax, mode="wrap", origin=-1)
                    case(fn, {"input": x}, kw, getattr(ndi, fn)(x, **kw), family="minmax")
    x = rnd(rng, (6, 7), "uint8")
    for cv in [300, -5, 2.7]:
        for fn in ["minimum_filter", "maximum_filter"]:
            kw = dict(size=3, mode="constant", cval=cv)
            case(fn, {"input": x}, kw, getattr(ndi, fn)(x, **kw), family="minmax")
            fp = np.array([[1, 1, 0], [1, 1, 1], [0, 1, 1]], bool)
            kw = dict(mode="constant", cval=cv)
            case(fn, {"input": x, "footprint": fp}, kw, getattr(ndi, fn)(x, footprint=fp, **kw), family="minmax")

    # ---------------------------------------------------------------- binary
    for shape in [(17,), (12, 13), (7, 8, 9)]:
        nd = len(shape)
        for dens in [0.3, 0.7]:
            x = rng.random(shape) > dens
            mask = rng.random(shape) > 0.3
            structs = [None, np.ones((3,) * nd, bool), rng.random((3,) * nd) > 0.4, rng.random((2,) * nd) > 0.3]
            for st in structs:
                if st is not None and not st.any():
                    continue
                for it in [1, 2, 3]:
                    for bv in [0, 1]:
                        for m in [None, mask]:
                            for fn in ["binary_erosion", "binary_dilation"]:
                                arrs = {"input": x}
                                kw = dict(iterations=it, border_value=bv)
                                skw = dict(kw)
                                if st is not None:
                                    arrs["structure"] = st
                                    skw["structure"] = st
                                if m is not None:
                                    arrs["mask"] = m
                                    skw["mask"] = m
                                case(fn, arrs, kw, getattr(ndi, fn)(x, **skw), family="binary")
            for fn in ["binary_erosion", "binary_dilation"]:
                kw = dict(iterations=0)
                case(fn, {"input": x}, kw, getattr(ndi, fn)(x, **kw), family="binary")
                case(fn, {"input": x, "mask": mask}, kw, getattr(ndi, fn)(x, mask=mask, **kw), family="binary")
                kw = dict(origin=-1)
                case(fn, {"input": x}, kw, getattr(ndi, fn)(x, **kw), family="binary")
        xi = rnd(rng, shape, "int16")
        case("binary_erosion", {"input": xi}, {}, ndi.binary_erosion(xi), family="binary")
        case("binary_dilation", {"input": xi}, {}, ndi.binary_dilation(xi), family="binary")
    for rank in [1, 2, 3]:
        for conn in range(1, rank + 1):
            case("generate_binary_structure", {}, dict(rank=rank, connectivity=conn),
                 ndi.generate_binary_structure(rank, conn), family="binary")

    # --------------------------------------------------------- interpolation
    for shape in [(20,), (12, 13), (6, 7, 8)]:
        nd = len(shape)
        for dt in ["float32", "float64", "uint8"]:
            x = rnd(rng, shape, dt)
            n = 600
            c = np.stack([rng.uniform(-2.5 * s, 3.5 * s, n) for s in shape])
            c[:, :60] = np.round(c[:, :60])
            c[:, 60:120] = np.round(c[:, 60:120] * 2) / 2
            # integral cval for integer images: a fractional one only probes
            # .5 rounding ties of (sum of weights) * cval, which are rounding noise
            cv = 1.5 if np.dtype(dt).kind == "f" else 3.0
            for order in [0, 1]:
                for mode in IMODES:
                    kw = dict(order=order, mode=mode, cval=cv, prefilter=False)
                    case("map_coordinates", {"input": x, "coordinates": c}, kw,
                         ndi.map_coordinates(x, c, **kw), tol=1e-12, family="interp")
            c32 = c.astype(np.float32)
            kw = dict(order=1, mode="constant", cval=0.0, prefilter=False)
            case("map_coordinates", {"input": x, "coordinates": c32}, kw, ndi.map_coordinates(x, c32, **kw),
                 tol=1e-12, family="interp")
            if nd > 1:
                th = 0.3
                M = np.eye(nd)
                M[0, 0] = np.cos(th); M[0, 1] = -np.sin(th); M[1, 0] = np.sin(th); M[1, 1] = np.cos(th)
                M *= 1.1
                off = rng.uniform(-2, 2, nd)
                for order in [0, 1]:
                    for mode in IMODES:
                        kw = dict(offset=off, order=order, mode=mode, cval=cv, prefilter=False)
                        case("affine_transform", {"input": x, "matrix": M}, kw,
                             ndi.affine_transform(x, M, **kw), tol=1e-12, family="interp")
                    oshape = tuple(s + 3 for s in shape)
                    kw = dict(offset=off, order=order, mode="nearest", output_shape=oshape, prefilter=False)
                    case("affine_transform", {"input": x, "matrix": M}, kw, ndi.affine_transform(x, M, **kw),
                         tol=1e-12, family="interp")
                H = np.eye(nd + 1)
                H[:nd, :nd] = M
                H[:nd, nd] = off
                kw = dict(order=1, mode="mirror", prefilter=False)
                case("affine_transform", {"input": x, "matrix": H}, kw, ndi.affine_transform(x, H, **kw),
                     tol=1e-12, family="interp")
                case("affine_transform", {"input": x, "matrix": H[:nd]}, kw, ndi.affine_transform(x, H[:nd], **kw),
                     tol=1e-12, family="interp")

    # ------------------------------------- down-scaled BASELINE.json replicas
    g = np.random.default_rng(0)
    xh = g.standard_normal((32, 32, 32)).astype(np.float32)
    case("uniform_filter", {"input": xh}, dict(size=5), ndi.uniform_filter(xh, size=5), family="baseline_H")
    case("gaussian_filter", {"input": xh}, dict(sigma=2), ndi.gaussian_filter(xh, sigma=2), family="baseline_B")
    case("uniform_filter", {"input": xh}, dict(size=9), ndi.uniform_filter(xh, size=9), family="baseline_E")
    xc = np.random.default_rng(1).integers(0, 256, size=(32, 32, 32)).astype(np.uint8)
    case("grey_erosion", {"input": xc}, dict(size=7), ndi.grey_erosion(xc, size=7), family="baseline_C")
    n = 24
    xd = g.standard_normal((n, n, n)).astype(np.float32)
    ang = np.deg2rad(7.0)
    R = np.array([[1, 0, 0], [0, np.cos(ang), -np.sin(ang)], [0, np.sin(ang), np.cos(ang)]])
    M = np.diag([1.02, 1.0, 1.0]) @ R
    ctr = (n - 1) / 2.0
    off = ctr - M @ np.array([ctr] * 3) + np.array([0.5, -1.25, 2.0])
    idx = np.indices((n, n, n)).reshape(3, -1).astype(np.float64)
    coords = (M @ idx + off[:, None]).reshape(3, n, n, n).astype(np.float32)
    kw = dict(order=1, mode="constant", cval=0.0, prefilter=False)
    case("map_coordinates", {"input": xd, "coordinates": coords}, kw, ndi.map_coordinates(xd, coords, **kw),
         tol=1e-6, family="baseline_D")
    kw = dict(offset=off, order=1, mode="constant", cval=0.0, prefilter=False)
    case("affine_transform", {"input": xd, "matrix": M}, kw, ndi.affine_transform(xd, M, **kw), tol=1e-6,
         family="baseline_D")

    for c in cases:
        for k, v in list(c["kwargs"].items()):
            if isinstance(v, np.ndarray):
                c["kwargs"][k] = v.tolist()
            elif isinstance(v, tuple):
                c["kwargs"][k] = list(v)
    meta = {"scipy": scipy.__version__, "numpy": np.__version__, "python": sys.version.split()[0],
            "n_cases": len(cases)}
    out = os.path.join(HERE, "scipy_fixtures.npz")
    np.savez_compressed(out, __cases__=np.array(json.dumps(cases)), __meta__=np.array(json.dumps(meta)), **arrays)
    print(meta, os.path.getsize(out) / 1e6, "MB")


if __name__ == "__main__":
    main()
