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
// rows per lane and step: as much of a range in flight at once as the registers of the loaded columns allow, no more than the range holds
int redo_rows_per_lane(int ncols, uint32_t range_rows) {
    const int most = ncols <= 1 ? 16 : (ncols == 2 ? 8 : 4);
    int rr = 4;
    while (rr * 2 <= most && static_cast<uint32_t>(rr) * 2 * 64 <= range_rows && range_rows % (static_cast<uint32_t>(rr) * 2 * 64) == 0) rr *= 2;
    return rr;
}
RedoFn redo_kernel(int ncols, int rr) {
    switch (ncols * 100 + rr) {
        case 4: return &fused_redo_waves<0, 4>;
        case 104: return &fused_redo_waves<1, 4>;
        case 108: return &fused_redo_waves<1, 8>;
        case 116: return &fused_redo_waves<1, 16>;
        case 204: return &fused_redo_waves<2, 4>;
        case 208: return &fused_redo_waves<2, 8>;
        case 304: return &fused_redo_waves<3, 4>;
        case 404: return &fused_redo_waves<4, 4>;
        default: return nullptr;
    }
}
}  // namespace rvk
