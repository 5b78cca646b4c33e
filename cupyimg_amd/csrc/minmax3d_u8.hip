// minmax3d_u8.hip -- fused separable 3-D min / max for uint8 volumes
// (grey_erosion / grey_dilation with a flat, full `size` structuring element:
// cupyimg/scipy/ndimage/morphology.py:769-884 -> filters.py:1385-1396, three
// K2 launches + two zero-filled ping-pong buffers in the reference).
//
// Round 1: the fused kernel is not built yet; the entry point reports
// MI_ERR_UNSUPPORTED and the host runs the three generic 1-D passes
// (mi_minmax1d), which are bit-exact.  See DESIGN.md "next".
#include "common.hpp"

using namespace mi;

extern "C" int mi_minmax3d_u8(const mi_array *in, const mi_array *out, const int size[3],
                              const int origin[3], const int mode[3], int cval, int is_max,
                              mi_stream stream)
{
    (void)in; (void)out; (void)size; (void)origin; (void)mode; (void)cval; (void)is_max; (void)stream;
    set_error("minmax3d_u8: fused kernel not built");
    return MI_ERR_UNSUPPORTED;
}
