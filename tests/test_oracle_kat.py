"""Pins the CPU oracle against the literal known-answer vectors of the
reference's own tests (tests/golden/kat_reference.json; transcribed by
tests/golden/make_kat_reference.py with file:line of every vector).  No GPU."""
import numpy as np

from _cases import call, load_kat
from oracle import ndimage as orc


def run_kat(mod, to_device=None):
    cases = load_kat()
    assert len(cases) > 1500
    for c in cases:
        arrs = {k: np.asarray(v) for k, v in c["arrays"].items()}
        if c["in_dtype"] and "input" in arrs:
            arrs["input"] = arrs["input"].astype(c["in_dtype"])
        kwargs = dict(c["kwargs"])
        if "output" in kwargs:
            kwargs["output"] = np.dtype(kwargs["output"])
        got = call(mod, c["func"], arrs, kwargs, to_device=to_device)
        exp = np.asarray(c["expected"])
        assert got.shape == exp.shape, (c["func"], c["src"], got.shape, exp.shape)
        if "output" in c["kwargs"]:
            assert got.dtype == np.dtype(c["kwargs"]["output"]), (c["func"], c["src"])
        err = np.abs(got.astype(np.float64) - exp.astype(np.float64))
        assert (err.max() if err.size else 0.0) < 1.5 * 10 ** -c["decimal"], (c["func"], c["kwargs"], c["src"], got, exp)
    return len(cases)


def test_oracle_reproduces_reference_known_answers():
    assert run_kat(orc) > 1500
