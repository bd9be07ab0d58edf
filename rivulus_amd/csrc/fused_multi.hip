#include "fused_table.hpp"
namespace rvk {
const FusedEntry *fused_entries_multi(size_t *n) {
    static const FusedEntry t[] = {
        RV_FUSED(2, 8, 1, 16, 0), RV_FUSED(2, 8, 1, 16, FF_VALIDITY), RV_FUSED(2, 8, 2, 16, FF_VALIDITY),
        RV_FUSED(2, 8, 1, 16, FF_PROJALL), RV_FUSED(2, 8, 1, 16, FF_VALIDITY | FF_PROJALL), RV_FUSED(2, 8, 2, 16, FF_VALIDITY | FF_PROJALL),
        RV_FUSED(2, 8, 2, 16, FF_PROJALL),  // (16 rows/lane variants measured slower: 2.5-2.6 ms against 1.87-1.96 on config 3)
        RV_FUSED(2, 8, 2, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL), RV_FUSED(2, 8, 1, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL),  // config 3
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
