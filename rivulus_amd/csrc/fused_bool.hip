#include "fused_table.hpp"
namespace rvk {
// Predicate = Boolean column(s) (RecordBatch::filter / FilterStream, record_batch.rs:221-243, stream.rs:116-163;
// also the later column groups of a wide batch and batches that carry String columns), 1..4 eight-byte
// columns compacted, with and without null bitmaps.  The first entry of a class is the default geometry.
const FusedEntry *fused_entries_bool(size_t *n) {
    static const FusedEntry t[] = {
        RV_FUSED(1, 16, 2, 16, FF_BOOL | FF_PROJALL), RV_FUSED(1, 16, 1, 16, FF_BOOL | FF_PROJALL),
        RV_FUSED(1, 16, 2, 16, FF_BOOL | FF_VALIDITY | FF_PROJALL), RV_FUSED(1, 16, 1, 16, FF_BOOL | FF_VALIDITY | FF_PROJALL),
        RV_FUSED(2, 8, 2, 16, FF_BOOL | FF_PROJALL), RV_FUSED(2, 8, 1, 16, FF_BOOL | FF_PROJALL),
        RV_FUSED(2, 8, 2, 16, FF_BOOL | FF_VALIDITY | FF_PROJALL), RV_FUSED(2, 8, 1, 16, FF_BOOL | FF_VALIDITY | FF_PROJALL),
        RV_FUSED(3, 4, 1, 16, FF_BOOL | FF_VALIDITY | FF_PROJALL), RV_FUSED(4, 4, 1, 16, FF_BOOL | FF_VALIDITY | FF_PROJALL),
        // three and four columns without a null bitmap compacted by a Boolean column: the later column groups of a wide
        // projection (`selection is true -> [c4 .. c7]`, query.hip) -- the eager Filter keeps every column (plan.rs:132-147); without
        // these the launch fell to the every-feature instantiation (1.64 ms per 2e8 rows of four columns against 1.19)
        RV_FUSED(3, 8, 1, 16, FF_BOOL | FF_PROJALL), RV_FUSED(4, 4, 1, 16, FF_BOOL | FF_PROJALL),
        // Boolean columns PROJECTED: compacted inside the pass as bit streams (lane form: a PEXT per 64-row word), with a
        // Boolean or a value predicate
        RV_FUSED(1, 16, 1, 16, FF_BOOL | FF_XS | FF_PROJALL), RV_FUSED(1, 16, 1, 16, FF_BOOL | FF_XS | FF_VALIDITY | FF_PROJALL),
        RV_FUSED(2, 8, 1, 16, FF_BOOL | FF_XS | FF_VALIDITY | FF_PROJALL),
        RV_FUSED(1, 16, 1, 16, FF_XS | FF_PROJALL), RV_FUSED(1, 16, 1, 16, FF_XS | FF_VALIDITY | FF_PROJALL),
        RV_FUSED(2, 8, 1, 16, FF_XS | FF_VALIDITY | FF_PROJALL),
    };
    *n = sizeof(t) / sizeof(t[0]);
    return t;
}
}  // namespace rvk
