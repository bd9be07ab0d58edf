#include "fused_table.hpp"
namespace rvk {
const FusedEntry *fused_entries_lean1(size_t *n) {
    static const FusedEntry t[] = {
        RV_FUSED(1, 16, 1, 16, 0), RV_FUSED(1, 16, 2, 16, 0), RV_FUSED(1, 8, 1, 16, 0), RV_FUSED(1, 8, 2, 16, 0),
        RV_FUSED(1, 16, 1, 8, 0),  RV_FUSED(1, 16, 2, 8, 0),  RV_FUSED(1, 32, 1, 8, 0), RV_FUSED(1, 32, 2, 8, 0),
    };
    *n = sizeof(t) / sizeof(t[0]);
    return t;
}
}  // namespace rvk
