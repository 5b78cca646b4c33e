// median3d_f64.hip -- rank filters of the full 3 x 3 x 3 window of a float64 volume, every rank (median3d_impl.hpp: KeyOps64)
#include "median3d_impl.hpp"

namespace mi {
MI_RANK27_INST(double, true)
}  // namespace mi
