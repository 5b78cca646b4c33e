"""skimage look-alikes that sit directly on the ndimage hot path (SURVEY.md
section 8a row a15): grey / binary erosion + dilation, gaussian, warp.
Argument massaging only -- all device work happens in cupyimg_amd.scipy.ndimage."""
from . import filters, morphology, transform  # noqa: F401
