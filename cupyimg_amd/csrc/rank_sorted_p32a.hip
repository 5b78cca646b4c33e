// rank_sorted_p32a.hip -- explicit instantiations of the sorting-network rank kernel (rank_sorted.hpp)
#include "rank_sorted.hpp"

namespace mi {
MI_RANK_SORTED_INST(float, float, 32);
MI_RANK_SORTED_INST(uint8_t, float, 32);
MI_RANK_SORTED_INST(int8_t, float, 32);
MI_RANK_SORTED_INST(uint16_t, float, 32);
}  // namespace mi
