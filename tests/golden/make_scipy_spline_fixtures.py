"""Generates tests/golden/scipy_spline_fixtures.npz from SciPy: B-spline
prefilter and interpolation of order 2-5 (spline_filter, spline_filter1d,
map_coordinates, affine_transform, shift, zoom, rotate), the arithmetic truth
the reference's tests compare against (tests/test_interpolation.py:24-243,
tests/test_spline_vs_ndimage.py of the reference call scipy.ndimage the same way).

Run in the authoring container:   python tests/golden/make_scipy_spline_fixtures.py
Same file format as scipy_fixtures.npz; inputs, arguments and SciPy's outputs only.
"""
import json
import os
import sys
import warnings

import numpy as np
import scipy
import scipy.ndimage as ndi

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_scipy_fixtures as base  # noqa: E402  (put / case / rnd helpers and their tables)

HERE = os.path.dirname(os.path.abspath(__file__))
IMODES = base.IMODES


def main():
    warnings.simplefilter("ignore")
    rng = np.random.default_rng(20260101)
    case, rnd = base.case, base.rnd
    for shape in [(19,), (11, 13), (6, 7, 8), (1, 9)]:
        nd = len(shape)
        for dt in ["float64", "float32", "uint8"]:
            x = rnd(rng, shape, dt)
            cv = 1.5 if np.dtype(dt).kind == "f" else 3.0
            n = 200
            c = np.stack([rng.uniform(-2.5 * s - 3, 3.5 * s + 3, n) for s in shape])
            c[:, :30] = np.round(c[:, :30])
            c[:, 30:60] = np.round(c[:, 30:60] * 2) / 2
            for order in [2, 3, 4, 5]:
                for mode in IMODES:
                    if dt == "float64":
                        kw = dict(order=order, mode=mode)
                        case("spline_filter", {"input": x}, kw, ndi.spline_filter(x, **kw), tol=1e-12, family="spline_filter")
                        kw = dict(order=order, axis=nd - 1, mode=mode)
                        case("spline_filter1d", {"input": x}, kw, ndi.spline_filter1d(x, **kw), tol=1e-12,
                             family="spline_filter")
                    for pf in ([True, False] if dt == "float64" else [True]):
                        kw = dict(order=order, mode=mode, cval=cv, prefilter=pf)
                        case("map_coordinates", {"input": x, "coordinates": c}, kw, ndi.map_coordinates(x, c, **kw),
                             tol=1e-11 if dt == "float64" else (1e-6 if dt == "float32" else None), family="spline_map")
                    tol = 1e-11 if dt == "float64" else (1e-6 if dt == "float32" else None)
                    kw = dict(shift=[1.3, -0.6, 2.25][:nd], order=order, mode=mode, cval=cv)
                    case("shift", {"input": x}, kw, ndi.shift(x, **kw), tol=tol, family="spline_shift_zoom")
                    for gm in (False, True):
                        kw = dict(zoom=1.6, order=order, mode=mode, cval=cv, grid_mode=gm)
                        case("zoom", {"input": x}, kw, ndi.zoom(x, **kw), tol=tol, family="spline_shift_zoom")
                    if nd > 1 and dt != "uint8":
                        th = 0.3
                        M = np.eye(nd)
                        M[0, 0] = np.cos(th); M[0, 1] = -np.sin(th); M[1, 0] = np.sin(th); M[1, 1] = np.cos(th)
                        M *= 1.1
                        off = rng.uniform(-2, 2, nd)
                        kw = dict(offset=off, order=order, mode=mode, cval=cv)
                        case("affine_transform", {"input": x, "matrix": M}, kw, ndi.affine_transform(x, M, **kw),
                             tol=tol, family="spline_affine")
    for c in base.cases:
        for k, v in list(c["kwargs"].items()):
            if isinstance(v, np.ndarray):
                c["kwargs"][k] = v.tolist()
            elif isinstance(v, tuple):
                c["kwargs"][k] = list(v)
    meta = {"scipy": scipy.__version__, "numpy": np.__version__, "python": sys.version.split()[0],
            "n_cases": len(base.cases)}
    out = os.path.join(HERE, "scipy_spline_fixtures.npz")
    np.savez_compressed(out, __cases__=np.array(json.dumps(base.cases)), __meta__=np.array(json.dumps(meta)), **base.arrays)
    print(meta, os.path.getsize(out) / 1e6, "MB")


if __name__ == "__main__":
    main()
