#include "fused_table.hpp"
namespace rvk {
const FusedEntry *fused_entries_full(size_t *n) {
    static const FusedEntry t[] = {
        RV_FUSED(0, 16, 1, 16, FF_ALL), RV_FUSED(1, 8, 1, 16, FF_ALL), RV_FUSED(2, 4, 1, 16, FF_ALL),
        RV_FUSED(3, 4, 1, 16, FF_ALL),  RV_FUSED(4, 4, 1, 16, FF_ALL),
    };
    *n = sizeof(t) / sizeof(t[0]);
    return t;
}
RedoFn redo_kernel(int ncols, int rows_per_lane) {
#define RV_REDO(NC)                                                   \
    case NC:                                                          \
        return rows_per_lane == 8   ? &fused_redo_tiles<NC, 8>        \
               : rows_per_lane == 4 ? &fused_redo_tiles<NC, 4>        \
               : rows_per_lane == 2 ? &fused_redo_tiles<NC, 2>        \
                                    : nullptr;
    switch (ncols) {
        RV_REDO(0)
        RV_REDO(1)
        RV_REDO(2)
        RV_REDO(3)
        RV_REDO(4)
        default: return nullptr;
    }
#undef RV_REDO
}
}  // namespace rvk
