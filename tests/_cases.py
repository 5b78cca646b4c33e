"""Shared case runner: replays golden cases (tests/golden/*) against either the
CPU oracle (oracle.ndimage) or the HIP path (cupyimg_amd.scipy.ndimage)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_POSITIONAL = {
    "correlate1d": ["input", "weights"], "convolve1d": ["input", "weights"],
    "correlate": ["input", "weights"], "convolve": ["input", "weights"],
    "map_coordinates": ["input", "coordinates"], "affine_transform": ["input", "matrix"],
}
_DEVICE_ARRAYS = {"input", "mask", "coordinates"}


def load_scipy_fixtures(name="scipy_fixtures.npz"):
    z = np.load(os.path.join(GOLDEN, name))
    cases = json.loads(str(z["__cases__"]))
    meta = json.loads(str(z["__meta__"]))
    return z, cases, meta


def load_kat():
    with open(os.path.join(GOLDEN, "kat_reference.json")) as f:
        return json.load(f)["cases"]


def call(mod, func, arrs, kwargs, to_device=None):
    """Invoke mod.func with named arrays; device implementation gets device
    arrays for the volume-sized arguments and returns through .get()."""
    fn = getattr(mod, func)
    arrs = dict(arrs)
    kwargs = dict(kwargs)
    if to_device is not None:
        for k in list(arrs):
            if k in _DEVICE_ARRAYS:
                arrs[k] = to_device(arrs[k])
    if func == "generate_binary_structure":
        return np.asarray(fn(kwargs["rank"], kwargs["connectivity"]))
    args = [arrs.pop(n) for n in _POSITIONAL.get(func, ["input"])]
    kwargs.update(arrs)
    out = fn(*args, **kwargs)
    if hasattr(out, "get"):
        out = out.get()
    return np.asarray(out)


def compare(got, expected, tol, what=""):
    assert got.shape == expected.shape, "{}: shape {} != {}".format(what, got.shape, expected.shape)
    assert got.dtype == expected.dtype, "{}: dtype {} != {}".format(what, got.dtype, expected.dtype)
    if tol is None:
        if not np.array_equal(got, expected):
            bad = np.flatnonzero(got.ravel() != expected.ravel())
            raise AssertionError("{}: {} of {} elements differ (first at {}: got {} expected {})".format(
                what, bad.size, got.size, bad[0], got.ravel()[bad[0]], expected.ravel()[bad[0]]))
    else:
        g = got.astype(np.float64)
        e = expected.astype(np.float64)
        scale = max(float(np.abs(e).max()) if e.size else 0.0, 1.0)
        err = float(np.abs(g - e).max()) if e.size else 0.0
        assert err <= tol * scale, "{}: max abs err {:.3e} > {:.1e} * {:.3g}".format(what, err, tol, scale)


def maxnorm_rel(got, expected):
    """max|y - y_ref| / max|y_ref| -- the north-star tolerance metric."""
    e = expected.astype(np.float64)
    d = float(np.abs(got.astype(np.float64) - e).max()) if e.size else 0.0
    m = float(np.abs(e).max()) if e.size else 0.0
    return d / m if m > 0 else d
