// rank_sorted_p64b.hip -- explicit instantiations of the sorting-network rank kernel (rank_sorted.hpp)
#include "rank_sorted.hpp"

namespace mi {
MI_RANK_SORTED_INST(int8_t, float, 64);
MI_RANK_SORTED_INST(uint16_t, float, 64);
}  // namespace mi
