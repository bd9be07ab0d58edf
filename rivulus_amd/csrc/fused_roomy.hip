// Explicit instantiations of the fused kernel, part 7: "roomy" geometries of the 2..4-column shapes -- four rows per lane
// in 16-wave workgroups where the default table has none, and in 8-wave workgroups, whose LDS slots (one workgroup per CU, two
// stages) hold EVERY row of a wave: no wave range is left to the redo kernel, whatever the selectivity.  Taken when the context's
// last selectivity would crowd the default geometry's slots (fused_begin).  Slower than the defaults at 10 % (tiles of 2048
// to 4096 rows), 1.4 - 2.3 times faster than re-reading the dense tiles.
#include "fused_table.hpp"
namespace rvk {
const FusedEntry *fused_entries_roomy(size_t *n) {
    constexpr int V = FF_VALIDITY, A = FF_PROJALL, N = FF_NONULL;
    static const FusedEntry t[] = {
        // two columns: 16 waves x 256 rows x 18 bytes x two stages = 144 KiB -- every row fits already
        RV_FUSED(2, 4, 1, 16, 0), RV_FUSED(2, 4, 1, 16, V), RV_FUSED(2, 4, 1, 16, A), RV_FUSED(2, 4, 1, 16, V | A), RV_FUSED(2, 4, 1, 16, V | A | N),
        // three columns
        RV_FUSED(3, 4, 1, 16, 0), RV_FUSED(3, 4, 1, 16, V | A | N),
        RV_FUSED(3, 4, 1, 8, 0), RV_FUSED(3, 4, 1, 8, V), RV_FUSED(3, 4, 1, 8, A), RV_FUSED(3, 4, 1, 8, V | A), RV_FUSED(3, 4, 1, 8, V | A | N),
        // four columns
        RV_FUSED(4, 4, 1, 8, V), RV_FUSED(4, 4, 1, 8, A), RV_FUSED(4, 4, 1, 8, V | A),
    };
    *n = sizeof(t) / sizeof(t[0]);
    return t;
}
}  // namespace rvk
