// median3d_i.hip -- rank filters of the full 3 x 3 x 3 window of int8 / int32 / uint32 volumes, every rank (median3d_impl.hpp)
#include "median3d_impl.hpp"

namespace mi {
MI_RANK27_INST(int8_t, true)
MI_RANK27_INST(int32_t, true)
MI_RANK27_INST(uint32_t, true)
}  // namespace mi
