#include "fused_table.hpp"
namespace rvk {
// one 8-byte column, no null bitmap: BASELINE config 2.  The first entry of a (vec) class is the default.
const FusedEntry *fused_entries_lean1(size_t *n) {
    static const FusedEntry t[] = {
        RV_FUSED(1, 32, 2, 8, 0),  RV_FUSED(1, 32, 1, 8, 0),  RV_FUSED(1, 16, 2, 16, 0), RV_FUSED(1, 16, 1, 16, 0),
        RV_FUSED(1, 8, 2, 16, 0),  RV_FUSED(1, 8, 1, 16, 0),  RV_FUSED(1, 16, 2, 8, 0),  RV_FUSED(1, 16, 1, 8, 0),
        RV_FUSED(1, 16, 2, 16, FF_STAMP), RV_FUSED(1, 16, 1, 16, FF_STAMP), RV_FUSED(1, 16, 1, 8, FF_STAMP),  // diagnostic
    };
    *n = sizeof(t) / sizeof(t[0]);
    return t;
}
}  // namespace rvk
