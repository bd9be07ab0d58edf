#include "fused_table.hpp"
namespace rvk {
// Predicates with OR / NOT (rv_predicate::expr; BinaryOperator::Or expr.rs:28, BooleanArray::{or,not} boolean.rs:137-165):
// the literal list of a conjunctive normal form evaluated in the mask-major front end.  0..4 eight-byte columns,
// null bitmaps, Boolean predicate columns, selection bitmap on request.  Two accumulator mask arrays on top of the
// survive masks: 8 rows per lane keeps them in SGPRs.
const FusedEntry *fused_entries_expr(size_t *n) {
    constexpr int F = FF_VALIDITY | FF_BOOL | FF_SEL | FF_EXPR;
    static const FusedEntry t[] = {
        // 8-byte loads: the literals run in lane form (two VGPRs per mask set), which leaves room for more rows per lane
        RV_FUSED(0, 16, 1, 16, F), RV_FUSED(1, 16, 1, 16, F), RV_FUSED(2, 16, 1, 16, F), RV_FUSED(2, 8, 1, 16, F), RV_FUSED(3, 8, 1, 16, F),
        RV_FUSED(3, 4, 1, 16, F), RV_FUSED(4, 8, 1, 16, F), RV_FUSED(4, 4, 1, 16, F),
        RV_FUSED(1, 8, 2, 16, F), RV_FUSED(2, 8, 2, 16, F),  // 16-byte loads on request (option "vec" = 2)
        // every loaded column projected (FF_PROJALL) and, under strict null propagation, no null among the survivors
        // (FF_NONULL): the shapes `(a <op> x OR b <op> y) -> [a, b]` -- BASELINE config 3 with OR / NOT in it
        RV_FUSED(1, 16, 1, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL | FF_EXPR), RV_FUSED(2, 16, 1, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL | FF_EXPR),
        RV_FUSED(3, 8, 1, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL | FF_EXPR),
        RV_FUSED(1, 16, 1, 16, FF_PROJALL | FF_EXPR), RV_FUSED(2, 16, 1, 16, FF_PROJALL | FF_EXPR), RV_FUSED(3, 8, 1, 16, FF_PROJALL | FF_EXPR),
        // no Boolean predicate column, no selection bitmap: the same general form without their code
        // three columns at 12 rows per lane: `(f > 0.9 OR x < 50) AND y >= 100 -> [f, x]` 2.21 ms against 2.34 ms at 8 (5e8 rows, same
        // box; 16 rows per lane in 8-wave workgroups: 2.86 ms)
        RV_FUSED(2, 16, 1, 16, FF_VALIDITY | FF_EXPR), RV_FUSED(3, 12, 1, 16, FF_VALIDITY | FF_EXPR), RV_FUSED(3, 8, 1, 16, FF_VALIDITY | FF_EXPR),
        RV_FUSED(4, 4, 1, 16, FF_VALIDITY | FF_EXPR),
        // ... and no null among the survivors (strict propagation over every nullable column read) while some column is tested
        // but not projected: the value staging without the validity select
        RV_FUSED(2, 16, 1, 16, FF_VALIDITY | FF_NONULL | FF_EXPR), RV_FUSED(3, 12, 1, 16, FF_VALIDITY | FF_NONULL | FF_EXPR),
        RV_FUSED(3, 8, 1, 16, FF_VALIDITY | FF_NONULL | FF_EXPR), RV_FUSED(4, 4, 1, 16, FF_VALIDITY | FF_NONULL | FF_EXPR),
    };
    *n = sizeof(t) / sizeof(t[0]);
    return t;
}
}  // namespace rvk
