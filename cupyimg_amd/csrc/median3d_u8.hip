// median3d_u8.hip -- rank filters of the full 3 x 3 x 3 window of a uint8 volume, every rank (median3d_impl.hpp)
#include "median3d_impl.hpp"

namespace mi {
MI_RANK27_INST(uint8_t, true)
}  // namespace mi
