"""Zero-copy interop through __cuda_array_interface__ (version 3) with stream
ordering, and the per-stream arenas of the scratch pool."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ndi(gpu):
    from cupyimg_amd.scipy import ndimage
    return ndimage


def test_export_synchronises_or_names_the_default_stream(gpu):
    a = gpu.asarray(np.arange(12, dtype=np.float32).reshape(3, 4))
    cai = a.__cuda_array_interface__
    assert cai["version"] == 3 and cai["stream"] is None          # synchronised: nothing left to wait for
    gpu.core.EXPORT_SYNC = False
    try:
        cai = a.__cuda_array_interface__
        assert cai["stream"] == gpu.core.default_stream_handle() and cai["stream"] not in (0, None)
    finally:
        gpu.core.EXPORT_SYNC = True


def test_torch_round_trip_is_stream_ordered(gpu):
    """A tensor produced on torch's stream is filtered right away (no explicit sync) and the
    result is consumed by torch right away: import waits on the producer stream the tensor
    names, export names the library's stream (tests/helpers/torch_interop_check.py, run in a
    child process that imports torch first -- zero-copy interop needs ONE HIP runtime in the
    process, and this image's torch wheel ships its own)."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "torch_interop_check.py")
    proc = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    tail = "\n".join(proc.stdout.splitlines()[-15:])
    if "INTEROP_SKIP" in proc.stdout:
        pytest.skip(tail)
    assert proc.returncode == 0 and "INTEROP_OK" in proc.stdout, tail


def test_scratch_blocks_stay_with_their_stream(gpu):
    """Temporaries of a call on a user stream return to that stream's arena: a
    second stream never gets the block while the first may still be using it."""
    from cupyimg_amd import _lib
    lib = _lib.load()
    lib.mi_debug_pool_probe.argtypes = [ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    s1, s2 = gpu.Stream(), gpu.Stream()
    p1, p2, p3 = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    n = 3 << 20
    assert lib.mi_debug_pool_probe(n, s1.handle, ctypes.byref(p1)) == 0      # alloc on s1, free
    assert lib.mi_debug_pool_probe(n, s2.handle, ctypes.byref(p2)) == 0      # another stream: a different block
    assert lib.mi_debug_pool_probe(n, s1.handle, ctypes.byref(p3)) == 0      # same stream: the cached block
    assert p1.value != p2.value and p1.value == p3.value
