// rank_sorted_p32b.hip -- explicit instantiations of the sorting-network rank kernel (rank_sorted.hpp)
#include "rank_sorted.hpp"

namespace mi {
MI_RANK_SORTED_INST(int16_t, float, 32);
MI_RANK_SORTED_INST(double, double, 32);
MI_RANK_SORTED_INST(int32_t, double, 32);
MI_RANK_SORTED_INST(uint32_t, double, 32);
}  // namespace mi
