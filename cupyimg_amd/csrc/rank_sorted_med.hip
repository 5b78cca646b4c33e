// rank_sorted_med.hip -- medians of 25 (5 x 5) and 27 (3 x 3 x 3) samples: the sorting network of rank_sorted.hpp with the tap
// count and the rank fixed at compile time (padding folded, compare-exchanges that cannot reach the median removed)
#include "rank_sorted.hpp"

namespace mi {
MI_MEDIAN_SORTED_INST(float, float, 25);
MI_MEDIAN_SORTED_INST(uint8_t, float, 25);
MI_MEDIAN_SORTED_INST(uint16_t, float, 25);
MI_MEDIAN_SORTED_INST(int16_t, float, 25);
MI_MEDIAN_SORTED_INST(float, float, 27);
MI_MEDIAN_SORTED_INST(uint8_t, float, 27);
MI_MEDIAN_SORTED_INST(uint16_t, float, 27);
MI_MEDIAN_SORTED_INST(int16_t, float, 27);
}  // namespace mi
