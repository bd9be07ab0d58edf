#include "fused_table.hpp"
namespace rvk {
// one 8-byte column, no null bitmap.  FF_ONE_*: exactly one compare term (BASELINE config 2, `x > lit`).
// The first entry of a (flags, vec) class is the default geometry.
const FusedEntry *fused_entries_lean1(size_t *n) {
    static const FusedEntry t[] = {
        // measured on 1e9 Int64 rows at 10 % (tools/sweep.py): (16,2,16) 1.29 ms, (24,2,16) 1.29, (32,2,8) 1.40-1.48,
        // (32,1,8) 1.46, (16,2,8) 1.78, (8,2,16) 1.78
        RV_FUSED(1, 16, 2, 16, FF_ONE_I64), RV_FUSED(1, 16, 1, 16, FF_ONE_I64), RV_FUSED(1, 24, 2, 16, FF_ONE_I64),
        RV_FUSED(1, 32, 2, 8, FF_ONE_I64),  RV_FUSED(1, 32, 1, 8, FF_ONE_I64),  RV_FUSED(1, 16, 2, 8, FF_ONE_I64),
        RV_FUSED(1, 8, 2, 16, FF_ONE_I64),
        RV_FUSED(1, 16, 2, 16, FF_ONE_F64), RV_FUSED(1, 16, 1, 16, FF_ONE_F64), RV_FUSED(1, 8, 2, 16, FF_ONE_F64),
        // + selection bitmap (rv_eval_predicate, String projections, several column groups)
        RV_FUSED(1, 16, 2, 16, FF_ONE_I64 | FF_SEL), RV_FUSED(1, 16, 1, 16, FF_ONE_I64 | FF_SEL), RV_FUSED(1, 8, 2, 16, FF_ONE_I64 | FF_SEL),
        RV_FUSED(1, 16, 2, 16, FF_ONE_F64 | FF_SEL), RV_FUSED(1, 16, 1, 16, FF_ONE_F64 | FF_SEL),
        RV_FUSED(1, 16, 2, 16, 0), RV_FUSED(1, 16, 1, 16, 0), RV_FUSED(1, 32, 2, 8, 0), RV_FUSED(1, 8, 2, 16, 0),  // several terms
        RV_FUSED(1, 16, 2, 16, FF_PROJALL), RV_FUSED(1, 16, 1, 16, FF_PROJALL),
        // four rows per lane: three stages of EVERY row of a wave fit the LDS (16 x 256 x 8 B x 3 = 96 KiB) -- the geometry of
        // dense selections (fused_begin walks down to it from the context's last selectivity)
        RV_FUSED(1, 4, 2, 16, FF_ONE_I64), RV_FUSED(1, 4, 2, 16, FF_ONE_F64), RV_FUSED(1, 4, 2, 16, FF_ONE_I64 | FF_SEL), RV_FUSED(1, 4, 2, 16, FF_ONE_F64 | FF_SEL),
        RV_FUSED(1, 4, 2, 16, 0), RV_FUSED(1, 8, 2, 16, FF_PROJALL), RV_FUSED(1, 4, 2, 16, FF_PROJALL),
        RV_FUSED(1, 16, 2, 16, FF_ONE_I64 | FF_STAMP), RV_FUSED(1, 32, 2, 8, FF_ONE_I64 | FF_STAMP),  // diagnostic (option "stamp")
    };
    *n = sizeof(t) / sizeof(t[0]);
    return t;
}
}  // namespace rvk
