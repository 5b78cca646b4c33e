"""HIP path vs the committed SciPy 1.15.3 fixtures, through the C-ABI
(cupyimg_amd.scipy.ndimage -> libmi355img.so).  GPU only.

Tolerances (stated, per BASELINE.json north_star):
  * integer / bool outputs and all morphology: bit-exact;
  * correlate1d / convolve1d / correlate / convolve: bit-exact (the kernels
    accumulate in double in SciPy's summation order, built without FMA
    contraction);
  * uniform / gaussian with float output: max|y - y_ref| <= 1e-6 * max|y_ref|
    for float32 (the fused kernel computes in float32), 1e-12 for float64;
  * interpolation: 1e-12 for float64 outputs (double arithmetic), 2e-6 * max(1, max|ref|) for
    float32 volumes (float32 weights / accumulation in interp_fast.hip).
"""
import numpy as np
import pytest

from _cases import call, compare, load_scipy_fixtures, maxnorm_rel

pytestmark = pytest.mark.gpu

Z, CASES, META = load_scipy_fixtures()
FAMILIES = sorted({c["family"] for c in CASES})


def _check(c, got, expected, inp_dtype=None):
    what = "case {} {} {}".format(c["id"], c["func"], c["kwargs"])
    fam = c["family"]
    assert got.shape == expected.shape and got.dtype == expected.dtype, what
    if expected.dtype.kind in "iub":
        compare(got, expected, None, what)
    elif fam in ("uniform", "gaussian") or fam.startswith("baseline_"):
        if fam == "baseline_D":
            compare(got, expected, 2e-6, what)
        else:
            lim = 1e-6 if expected.dtype == np.float32 else 1e-12
            r = maxnorm_rel(got, expected)
            assert r <= lim, "{}: max-norm rel err {:.3e} > {:.0e}".format(what, r, lim)
    elif fam in ("corr1d", "corrnd"):
        if c["func"] in ("correlate1d", "convolve1d") and inp_dtype == np.float32:
            # reference default for the 1-D entry points is dtype_mode="float":
            # float32 accumulation for float32 input (filters.py:223,297)
            r = maxnorm_rel(got, expected)
            assert r <= 1e-6, "{}: max-norm rel err {:.3e}".format(what, r)
        else:
            compare(got, expected, None, what)
    elif fam == "interp":
        # float32 volumes take the float32-weight kernels (interp_fast.hip): stated tolerance 2e-6
        compare(got, expected, 2e-6 if expected.dtype == np.float32 else c["tol"], what)
    else:
        compare(got, expected, None, what)


@pytest.mark.parametrize("family", FAMILIES)
def test_hip_matches_scipy_fixture(gpu, family):
    from cupyimg_amd.scipy import ndimage as ndi
    n = 0
    for c in CASES:
        if c["family"] != family:
            continue
        arrs = {k: Z[v] for k, v in c["arrays"].items()}
        expected = Z[c["expected"]]
        got = call(ndi, c["func"], arrs, c["kwargs"], to_device=gpu.asarray)
        _check(c, got, expected, arrs["input"].dtype if "input" in arrs else None)
        n += 1
    assert n > 0


ZS, SCASES, SMETA = load_scipy_fixtures("scipy_spline_fixtures.npz")
SFAMILIES = sorted({c["family"] for c in SCASES})


@pytest.mark.parametrize("family", SFAMILIES)
def test_hip_matches_scipy_spline_fixture(gpu, family):
    """B-spline prefilter and interpolation of order 2-5 against SciPy's own
    outputs: 1e-11 for float64, 1e-6 for float32 results, exact for uint8."""
    import warnings

    from cupyimg_amd.scipy import ndimage as ndi
    n = 0
    for c in SCASES:
        if c["family"] != family:
            continue
        arrs = {k: ZS[v] for k, v in c["arrays"].items()}
        expected = ZS[c["expected"]]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = call(ndi, c["func"], arrs, c["kwargs"], to_device=gpu.asarray)
        compare(got, expected, c["tol"], "case {} {} {}".format(c["id"], c["func"], c["kwargs"]))
        n += 1
    assert n > 0
