// rank_sorted_p64d.hip -- explicit instantiations of the sorting-network rank kernel (rank_sorted.hpp)
#include "rank_sorted.hpp"

namespace mi {
MI_RANK_SORTED_INST(int32_t, double, 64);
MI_RANK_SORTED_INST(uint32_t, double, 64);
}  // namespace mi
