// rank_sorted_p128f.hip -- explicit instantiations of the sorting-network rank kernel (rank_sorted.hpp), 65..128 samples
// (5 x 5 x 5, 9 x 9 and 11 x 11 windows): 1792 compare-exchanges on 128 value registers, two and a half minutes of
// compile time each, hence one value type per file
#include "rank_sorted.hpp"

namespace mi {
MI_RANK_SORTED_INST(int32_t, double, 128);
}  // namespace mi
