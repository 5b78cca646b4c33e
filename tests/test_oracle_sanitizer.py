"""CPU sanitizer run of the oracle (sanitizers on the host side only): oracle/ndimage_oracle.c built
with -fsanitize=address,undefined (oracle/Makefile, liboracle_asan.so) and driven through the reference's known-answer
vectors and the SciPy fixtures in a child process with libasan preloaded.  An out-of-bounds tap, a signed overflow or a
misaligned access in the C restatement aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    out = subprocess.run(["gcc", "-print-file-name=" + name], stdout=subprocess.PIPE, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def test_oracle_under_asan_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan.so here")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"])
    env = dict(os.environ)
    preload = [asan] + ([_runtime("libubsan.so")] if _runtime("libubsan.so") else [])
    env.update(LD_PRELOAD=":".join(preload), ORACLE_LIB="liboracle_asan.so",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_oracle_kat.py"), os.path.join(ROOT, "tests", "test_oracle_scipy_fixtures.py")]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=ROOT, timeout=1500)
    tail = out.stdout[-4000:]
    assert out.returncode == 0, tail
    assert "AddressSanitizer" not in out.stdout and "runtime error" not in out.stdout, tail
    assert " passed" in out.stdout, tail
