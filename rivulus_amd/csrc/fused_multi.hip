#include "fused_table.hpp"
namespace rvk {
// 2..4 eight-byte columns.  The first entry of a (flags, load width) class is the default geometry
// (rows per lane, 16 waves): measured on config 3 (tools/sweep3.py) 12 rows 1.49 ms, 16 rows 1.50, 8 rows 1.58.
const FusedEntry *fused_entries_multi(size_t *n) {
    static const FusedEntry t[] = {
        // predicate column not projected, ... : per-column checks stay in the staging loop
        RV_FUSED(2, 8, 1, 16, 0), RV_FUSED(2, 8, 2, 16, 0), RV_FUSED(2, 8, 1, 16, FF_VALIDITY), RV_FUSED(2, 8, 2, 16, FF_VALIDITY),
        // every loaded column projected
        RV_FUSED(2, 12, 2, 16, FF_PROJALL), RV_FUSED(2, 8, 2, 16, FF_PROJALL), RV_FUSED(2, 8, 1, 16, FF_PROJALL),
        RV_FUSED(2, 12, 2, 16, FF_VALIDITY | FF_PROJALL), RV_FUSED(2, 8, 2, 16, FF_VALIDITY | FF_PROJALL), RV_FUSED(2, 8, 1, 16, FF_VALIDITY | FF_PROJALL),
        // ... and no null can survive (BASELINE config 3)
        // 8-byte loads first choice here (fused_begin): with them a slot is one word of every bitmap and the whole
        // predicate runs in lane form (fused_kernel.hpp); measured on config 3, 5e8 rows, same box: (16,1,16) 1.435 ms,
        // (12,2,16) 1.562, (16,1,8) 1.497, (16,2,8) 1.629, (24,1,8) 1.627
        RV_FUSED(2, 16, 1, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL), RV_FUSED(2, 12, 1, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL),
        RV_FUSED(2, 8, 1, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL),
        RV_FUSED(2, 12, 2, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL), RV_FUSED(2, 16, 2, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL),
        RV_FUSED(2, 8, 2, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL),
        RV_FUSED(2, 8, 1, 16, FF_VALIDITY | FF_PROJALL | FF_STAMP),  // diagnostic (option "stamp")
        RV_FUSED(3, 4, 1, 16, FF_VALIDITY), RV_FUSED(4, 4, 1, 16, FF_VALIDITY),
        RV_FUSED(3, 8, 2, 16, FF_PROJALL), RV_FUSED(3, 8, 2, 16, FF_VALIDITY | FF_PROJALL), RV_FUSED(3, 8, 2, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL),
        RV_FUSED(3, 8, 1, 16, FF_PROJALL), RV_FUSED(3, 8, 1, 16, FF_VALIDITY | FF_PROJALL), RV_FUSED(3, 8, 1, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL),
        RV_FUSED(3, 4, 1, 16, FF_PROJALL), RV_FUSED(4, 4, 1, 16, FF_PROJALL), RV_FUSED(4, 4, 1, 16, FF_VALIDITY | FF_PROJALL),
    };
    *n = sizeof(t) / sizeof(t[0]);
    return t;
}
}  // namespace rvk
