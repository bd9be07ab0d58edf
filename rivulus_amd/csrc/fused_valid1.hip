#include "fused_table.hpp"
namespace rvk {
const FusedEntry *fused_entries_valid1(size_t *n) {
    static const FusedEntry t[] = {
        RV_FUSED(1, 16, 2, 16, FF_VALIDITY), RV_FUSED(1, 16, 1, 16, FF_VALIDITY),
        RV_FUSED(1, 8, 2, 16, FF_VALIDITY),  RV_FUSED(1, 8, 1, 16, FF_VALIDITY),
        // the column is projected with its bitmap (the usual shape): no per-column checks in the staging loop
        RV_FUSED(1, 16, 2, 16, FF_VALIDITY | FF_PROJALL), RV_FUSED(1, 16, 1, 16, FF_VALIDITY | FF_PROJALL),
        RV_FUSED(1, 8, 2, 16, FF_VALIDITY | FF_PROJALL),
        RV_FUSED(1, 16, 2, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL), RV_FUSED(1, 16, 1, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL),
        // four rows per lane: the slots hold every row of a wave (dense selections; 9-byte rows: two stages)
        RV_FUSED(1, 4, 2, 16, FF_VALIDITY), RV_FUSED(1, 4, 1, 16, FF_VALIDITY), RV_FUSED(1, 4, 2, 16, FF_VALIDITY | FF_PROJALL), RV_FUSED(1, 4, 1, 16, FF_VALIDITY | FF_PROJALL),
        RV_FUSED(1, 8, 2, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL), RV_FUSED(1, 8, 1, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL),
        RV_FUSED(1, 4, 2, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL), RV_FUSED(1, 4, 1, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL),
    };
    *n = sizeof(t) / sizeof(t[0]);
    return t;
}
}  // namespace rvk
