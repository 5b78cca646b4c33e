// rank_sorted_p16.hip -- explicit instantiations of the sorting-network rank kernel (rank_sorted.hpp)
#include "rank_sorted.hpp"

namespace mi {
MI_RANK_SORTED_INST(float, float, 16);
MI_RANK_SORTED_INST(uint8_t, float, 16);
MI_RANK_SORTED_INST(int8_t, float, 16);
MI_RANK_SORTED_INST(uint16_t, float, 16);
MI_RANK_SORTED_INST(int16_t, float, 16);
MI_RANK_SORTED_INST(double, double, 16);
MI_RANK_SORTED_INST(int32_t, double, 16);
MI_RANK_SORTED_INST(uint32_t, double, 16);
}  // namespace mi
