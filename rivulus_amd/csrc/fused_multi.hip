#include "fused_table.hpp"
namespace rvk {
const FusedEntry *fused_entries_multi(size_t *n) {
    static const FusedEntry t[] = {
        RV_FUSED(2, 8, 1, 16, 0), RV_FUSED(2, 8, 1, 16, FF_VALIDITY), RV_FUSED(2, 16, 1, 16, FF_VALIDITY), RV_FUSED(2, 8, 2, 16, FF_VALIDITY),
        RV_FUSED(2, 16, 2, 16, FF_VALIDITY), RV_FUSED(2, 16, 1, 8, FF_VALIDITY), RV_FUSED(2, 16, 2, 8, FF_VALIDITY),
        RV_FUSED(3, 4, 1, 16, FF_VALIDITY), RV_FUSED(4, 4, 1, 16, FF_VALIDITY),
    };
    *n = sizeof(t) / sizeof(t[0]);
    return t;
}
}  // namespace rvk
