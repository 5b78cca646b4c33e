"""Pins the CPU oracle against the committed SciPy 1.15.3 fixtures
(tests/golden/scipy_fixtures.npz, generator: tests/golden/make_scipy_fixtures.py).
Runs without a GPU."""
import numpy as np
import pytest

from _cases import call, compare, load_scipy_fixtures
from oracle import ndimage as orc

Z, CASES, META = load_scipy_fixtures()
FAMILIES = sorted({c["family"] for c in CASES})


def _tol_for_oracle(c, expected):
    if c["tol"] is not None:
        return c["tol"]
    # gaussian derivative weights are built with a different (equivalent)
    # polynomial recurrence: last-bit differences in the weights
    if c["func"].startswith("gaussian") and expected.dtype.kind == "f":
        order = c["kwargs"].get("order", 0)
        if np.any(np.asarray(order) > 0):
            return 1e-12 if expected.dtype == np.float64 else 1e-6
    return None


@pytest.mark.parametrize("family", FAMILIES)
def test_oracle_matches_scipy_fixture(family):
    n = 0
    for c in CASES:
        if c["family"] != family:
            continue
        arrs = {k: Z[v] for k, v in c["arrays"].items()}
        expected = Z[c["expected"]]
        got = call(orc, c["func"], arrs, c["kwargs"])
        compare(got, expected, _tol_for_oracle(c, expected), "case {} {} {}".format(c["id"], c["func"], c["kwargs"]))
        n += 1
    assert n > 0


ZS, SCASES, SMETA = load_scipy_fixtures("scipy_spline_fixtures.npz")
SFAMILIES = sorted({c["family"] for c in SCASES})


@pytest.mark.parametrize("family", SFAMILIES)
def test_oracle_matches_scipy_spline_fixture(family):
    """B-spline prefilter / orders 2-5 (tests/golden/make_scipy_spline_fixtures.py)"""
    n = 0
    for c in SCASES:
        if c["family"] != family:
            continue
        arrs = {k: ZS[v] for k, v in c["arrays"].items()}
        expected = ZS[c["expected"]]
        got = call(orc, c["func"], arrs, c["kwargs"])
        compare(got, expected, c["tol"], "case {} {} {}".format(c["id"], c["func"], c["kwargs"]))
        n += 1
    assert n > 0
    assert SMETA["scipy"] == "1.15.3"


def test_fixture_metadata():
    assert META["scipy"] == "1.15.3"
    assert META["n_cases"] == len(CASES)
