// rank_sorted_p64c.hip -- explicit instantiations of the sorting-network rank kernel (rank_sorted.hpp)
#include "rank_sorted.hpp"

namespace mi {
MI_RANK_SORTED_INST(int16_t, float, 64);
MI_RANK_SORTED_INST(double, double, 64);
}  // namespace mi
