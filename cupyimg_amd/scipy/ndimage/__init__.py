"""scipy.ndimage-compatible API on device arrays.

Mirrors cupyimg/scipy/ndimage/__init__.py:1-16 for the filtering hot path
(filters, morphology, interpolation).
"""
from .filters import *  # noqa: F401,F403
from .morphology import *  # noqa: F401,F403
from .interpolation import *  # noqa: F401,F403
