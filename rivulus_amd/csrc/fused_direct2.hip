// More instantiations of the direct kernel (direct_kernel.hpp; see fused_direct.hip): two to four predicate columns, and
// launches whose predicate reads Boolean columns only.
#include "direct_kernel.hpp"
#include "fused_table.hpp"
namespace rvk {
#define RV_DIRECT(NP, NQ, R, W, F) DirectEntry{NP, NQ, R, W, F, &fused_direct_compact<NP, NQ, R, W, F, 4>}
#define RV_DIRECT3(NP, NQ, R, W) RV_DIRECT(NP, NQ, R, W, 0), RV_DIRECT(NP, NQ, R, W, FF_VALIDITY), RV_DIRECT(NP, NQ, R, W, FF_VALIDITY | FF_BOOL)
const DirectEntry *direct_entries_b(size_t *n) {
    static const DirectEntry t[] = {
        RV_DIRECT3(2, 1, 6, 8), RV_DIRECT3(2, 2, 4, 8), RV_DIRECT3(3, 0, 4, 8), RV_DIRECT3(3, 1, 4, 8), RV_DIRECT3(4, 0, 4, 8),
        RV_DIRECT(0, 1, 16, 8, FF_VALIDITY | FF_BOOL), RV_DIRECT(0, 2, 8, 8, FF_VALIDITY | FF_BOOL), RV_DIRECT(0, 3, 4, 8, FF_VALIDITY | FF_BOOL), RV_DIRECT(0, 4, 4, 8, FF_VALIDITY | FF_BOOL),
        // a projected column keeps its nulls (see fused_direct.hip)
        RV_DIRECT(2, 1, 6, 8, FF_VALIDITY | FF_OUTVALID), RV_DIRECT(2, 1, 4, 8, FF_VALIDITY | FF_OUTVALID), RV_DIRECT(2, 2, 4, 8, FF_VALIDITY | FF_OUTVALID),
        // power-of-two wave ranges (see fused_direct.hip)
        RV_DIRECT3(2, 1, 4, 8),
    };
    *n = sizeof(t) / sizeof(t[0]);
    return t;
}
}  // namespace rvk
