"""A bounded run of the differential fuzzer (scripts/fuzz_vs_scipy.py): 4000 seeded random cases over the
filter, morphology, rank and interpolation families, random dtypes, shapes (unit axes, 256 k + 4 wide rows),
strided / transposed inputs, output dtypes, modes and origins, each compared with scipy.ndimage."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bounded_differential_fuzz(gpu):
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_vs_scipy.py"), "240", "11", "4000"],
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    tail = "\n".join(proc.stdout.splitlines()[-25:])
    assert proc.returncode == 0, tail
    m = re.search(r"cases (\d+), refused on both sides \d+, failures (\d+)", proc.stdout)
    assert m, tail
    assert int(m.group(1)) >= 4000 and int(m.group(2)) == 0, tail
