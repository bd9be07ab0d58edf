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
RedoFn redo_kernel(int ncols) {
    switch (ncols) {
        case 0: return &fused_redo_tiles<0>;
        case 1: return &fused_redo_tiles<1>;
        case 2: return &fused_redo_tiles<2>;
        case 3: return &fused_redo_tiles<3>;
        case 4: return &fused_redo_tiles<4>;
        default: return nullptr;
    }
}
}  // namespace rvk
