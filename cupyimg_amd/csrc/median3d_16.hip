// median3d_16.hip -- rank filters of the full 3 x 3 x 3 window of 16-bit integer volumes, every rank (median3d_impl.hpp)
#include "median3d_impl.hpp"

namespace mi {
MI_RANK27_INST(uint16_t, true)
MI_RANK27_INST(int16_t, true)
}  // namespace mi
