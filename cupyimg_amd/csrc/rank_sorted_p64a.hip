// rank_sorted_p64a.hip -- explicit instantiations of the sorting-network rank kernel (rank_sorted.hpp)
#include "rank_sorted.hpp"

namespace mi {
MI_RANK_SORTED_INST(float, float, 64);
MI_RANK_SORTED_INST(uint8_t, float, 64);
}  // namespace mi
