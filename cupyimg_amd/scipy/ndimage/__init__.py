"""scipy.ndimage-compatible API on device arrays.

Mirrors cupyimg/scipy/ndimage/__init__.py:1-16 for the filtering hot path
(filters, morphology, interpolation).
"""
from .filters import *  # noqa: F401,F403
from .morphology import *  # noqa: F401,F403
from .interpolation import *  # noqa: F401,F403


def _wrap_float16():
    """Every public function that takes `output`: float16 images keep their dtype (_support.float16_aware)."""
    from . import filters, interpolation, morphology
    from ._support import float16_aware
    g = globals()
    for mod in (filters, morphology, interpolation):
        for name in getattr(mod, "__all__", ()):
            fn = getattr(mod, name)
            if callable(fn):
                wrapped = float16_aware(fn)
                g[name] = wrapped
                setattr(mod, name, wrapped)      # internal callers (gaussian_filter -> gaussian_filter1d) go through it too


_wrap_float16()
del _wrap_float16
