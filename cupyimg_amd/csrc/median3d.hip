// median3d.hip -- rank filters of the full 3 x 3 x 3 window of a volume (median3d_impl.hpp): float32; the knob.
#include "median3d_impl.hpp"

namespace mi {
static Knob g_median27{1};      // 0 = the per-voxel sorting network (rank_sorted.hpp) for the full 3 x 3 x 3 window too
int median27_enabled() { return g_median27; }
MI_RANK27_INST(float, true)
}  // namespace mi

extern "C" int mi_debug_set_median27(int k) { mi::g_median27 = k; return MI_OK; }
