"""cupyimg_amd -- MI355X-native n-D image filtering engine with the
scipy.ndimage / skimage drop-in API of mritools/cupyimg.

Python host code calls hand-written gfx950 HIP kernels through the ctypes
C-ABI in include/mi355img.h (libmi355img.so).  No CuPy, no PyTorch, no CPU
fallback: if the HIP library cannot be loaded, using the package fails.
"""
from . import core  # noqa: F401
from .core import (  # noqa: F401
    Event, Stream, array, arrays_differ, asarray, ascontiguousarray, asnumpy, device_count, device_name, empty,
    empty_like, free_all_blocks, full, get_device, is_available, ndarray, ones, pool_stats,
    set_device, shares_memory, synchronize, zeros, zeros_like,
)

__version__ = "0.1.0"


def convolve_separable(x, w, axes=None, **kwargs):
    """n-D convolution as separable convolve1d passes (cupyimg/_misc.py:39-77)"""
    from ._misc import convolve_separable as impl
    return impl(x, w, axes, **kwargs)


def last_kernel():
    """Name (with grid) of the kernel the last separable-filter call of this thread dispatched, as the library recorded
    it (`mi_debug_last_kernel`); a diagnostic for benchmarks and tests, empty before the first such call."""
    import ctypes
    from . import _lib
    buf = ctypes.create_string_buffer(256)
    fn = _lib.load().mi_debug_last_kernel
    fn.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    fn(buf, 256)
    return buf.value.decode()
