"""scipy look-alikes; only ``ndimage`` (the filtering hot path) is provided."""
from . import ndimage  # noqa: F401
from . import signal  # noqa: F401
